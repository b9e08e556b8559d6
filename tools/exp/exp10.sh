set -e
mkdir -p gpurun_out/r4
python tools/gpu_layers.py bf16 8 512 large > gpurun_out/r4/exp10_layers_large.txt 2>&1
python tools/gpu_layers.py fp16 32 256 base > gpurun_out/r4/exp10_layers_base.txt 2>&1 || true
