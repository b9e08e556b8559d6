set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python tools/gpu_kernel_sweep.py pw_gemm -- "" "gemm_bm256=1" "gemm_bm256=600" "" > gpurun_out/r4/exp25_sweep.txt 2>&1
grep -E "^==|128, 128|256, 128" gpurun_out/r4/exp25_sweep.txt | head -60
timeout -k 10 300 python tools/gpu_kernel_sweep.py pw_gemm bf16 8 512 large -- "" "gemm_bm256=1" > gpurun_out/r4/exp25_sweep_large.txt 2>&1
grep -E "^==|128, 128|256, 128" gpurun_out/r4/exp25_sweep_large.txt | head -60
