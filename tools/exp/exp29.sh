set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -m gpu -k "upconv64 or conv3x3_entry" > gpurun_out/r4/exp29_t0.txt 2>&1 || { tail -40 gpurun_out/r4/exp29_t0.txt; exit 1; }
tail -3 gpurun_out/r4/exp29_t0.txt
timeout -k 10 300 python tools/gpu_kernel_sweep.py conv -- "upconv64=1" "upconv64=0" "upconv64=3" "upconv64=5" "upconv64=9" "upconv64=1" > gpurun_out/r4/exp29_sweep.txt 2>&1
grep -E "^==|C=64 128" gpurun_out/r4/exp29_sweep.txt
timeout -k 10 200 python tools/gpu_knobs.py 32 "upconv64=1" "upconv64=0" "upconv64=1" "upconv64=0" > gpurun_out/r4/exp29_knobs.txt 2>&1
tail -5 gpurun_out/r4/exp29_knobs.txt
