set -e
mkdir -p gpurun_out/r4
python tools/gpu_knobs.py 32 "" "nt_mask=33" "nt_mask=65" "nt_mask=97" "nt_mask=101" > gpurun_out/r4/exp16_step.txt 2>&1
python tools/gpu_layers.py fp16 32 256 small nt_mask=97 > gpurun_out/r4/exp16_layers_nt97.txt 2>&1
python tools/gpu_layers.py fp16 32 256 small > gpurun_out/r4/exp16_layers_def.txt 2>&1
