timeout -k 10 200 python tools/exp/dbg_trainstep.py bf16 dirty 2>&1 | grep -v worst | tail -20
timeout -k 10 200 python tools/exp/dbg_trainstep.py fp32 2>&1 | grep -v worst | tail -20
timeout -k 10 200 python tools/exp/dbg_trainstep.py fp32 dirty 2>&1 | grep -v worst | tail -20
