set -e
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu -k "conv3x3 or downsample or upsample or unet_forward or enhance_small or full_size or operator_shapes or batch_equals" > gpurun_out/r4/exp21_tests.txt 2>&1 || true
tail -5 gpurun_out/r4/exp21_tests.txt
python tools/gpu_kernel_sweep.py conv3x3 fp16 32 256 small -- "" "conv_th16=0" "" "conv_th16=0" > gpurun_out/r4/exp21_sweep.txt 2>&1
python tools/gpu_knobs.py 32 "" "conv_th16=0" > gpurun_out/r4/exp21_step.txt 2>&1
