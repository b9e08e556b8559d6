timeout -k 10 500 python tools/gpu_ddp_check.py 2>&1 | grep -v amdgpu.ids | tail -8
