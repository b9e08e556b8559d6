set -e
mkdir -p gpurun_out/r4
python tools/gpu_layers.py fp16 16 256 small > gpurun_out/r4/exp6_layers_b16.txt 2>&1
python tools/gpu_layers.py fp16 32 256 small > gpurun_out/r4/exp6_layers_b32.txt 2>&1
python tools/gpu_knobs.py 32 "" "enhance_split=1" "enhance_split=3" "enhance_split=4" > gpurun_out/r4/exp6_split.txt 2>&1
