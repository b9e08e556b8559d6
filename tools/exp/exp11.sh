set -e
mkdir -p gpurun_out/r4
python tools/gpu_kernel_sweep.py _conv_ fp16 32 256 small -- "" > gpurun_out/r4/exp11_heads.txt 2>&1
python tools/gpu_knobs.py 32 "" > gpurun_out/r4/exp11_step.txt 2>&1
python -m pytest tests -x -q -m gpu -k "enhance or unet_forward or full_size or scheduler_step or smoke or final or init" > gpurun_out/r4/exp11_tests.txt 2>&1 || true
tail -3 gpurun_out/r4/exp11_tests.txt
