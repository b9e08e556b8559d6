set -e
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu -k "inverted_residual or unet_forward or enhance_small or full_size or batch_equals" > gpurun_out/r4/exp15_tests.txt 2>&1 || true
tail -5 gpurun_out/r4/exp15_tests.txt
python tools/gpu_knobs.py 32 "" "sgemm=0" > gpurun_out/r4/exp15_step.txt 2>&1
python tools/gpu_knobs.py 1 "" "sgemm=0" >> gpurun_out/r4/exp15_step.txt 2>&1
python tools/gpu_kernel_sweep.py gemm fp16 32 256 small -- "" "sgemm=0" > gpurun_out/r4/exp15_sweep.txt 2>&1
