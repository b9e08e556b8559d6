set -e
mkdir -p gpurun_out/r4
python tools/gpu_tune.py gemm_stamp > gpurun_out/r4/exp13_gemm_stamp.txt 2>&1 || true
python tools/gpu_tune.py gemm > gpurun_out/r4/exp13_gemm.txt 2>&1 || true
