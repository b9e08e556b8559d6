set -e
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu -k "squeeze or inverted_residual or unet_forward or enhance_small or full_size or batch_equals or large or se_mlp" > gpurun_out/r4/exp18_tests.txt 2>&1 || true
tail -5 gpurun_out/r4/exp18_tests.txt
python tools/gpu_knobs.py 32 "" "se_mfma=0" > gpurun_out/r4/exp18_step.txt 2>&1
python tools/gpu_knobs.py 1 "" "se_mfma=0" >> gpurun_out/r4/exp18_step.txt 2>&1
python tools/gpu_kernel_sweep.py se_ fp16 32 256 small -- "" "se_mfma=0" > gpurun_out/r4/exp18_sweep.txt 2>&1
