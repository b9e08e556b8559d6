set -e
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "se_gate_from or do_not_change_a_bit" > gpurun_out/r4/exp36_t0.txt 2>&1 || { tail -40 gpurun_out/r4/exp36_t0.txt; exit 1; }
tail -3 gpurun_out/r4/exp36_t0.txt
timeout -k 10 400 python tools/gpu_knobs.py 32 "" "se_tail=0" "" "se_tail=0" "" "se_tail=0" > gpurun_out/r4/exp36_b32.txt 2>&1; grep "B=" gpurun_out/r4/exp36_b32.txt
timeout -k 10 400 python tools/gpu_knobs.py 1 "" "se_tail=0" "" "se_tail=0" > gpurun_out/r4/exp36_b1.txt 2>&1; grep "B=" gpurun_out/r4/exp36_b1.txt
