set -e
mkdir -p gpurun_out/r4
for cfg in "256 32 32" "128 64 32" "64 128 32" "256 32 16"; do python tools/gpu_conv_stamp.py $cfg >> gpurun_out/r4/exp7_conv_stamps.txt 2>&1; done
