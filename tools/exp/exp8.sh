set -e
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/exp8_conv.txt
for cfg in "256 32 32" "128 64 32" "64 128 32" "256 32 16"; do python tools/gpu_conv_stamp.py $cfg >> gpurun_out/r4/exp8_conv.txt 2>&1; done
python -m pytest tests -x -q -m gpu -k "conv3x3 or downsample or upsample or unet_forward or full_size" > gpurun_out/r4/exp8_tests.txt 2>&1 || true
tail -3 gpurun_out/r4/exp8_tests.txt
python tools/gpu_knobs.py 32 "" > gpurun_out/r4/exp8_step.txt 2>&1
