set -e
mkdir -p gpurun_out/r4
python tools/gpu_knobs.py 32 "" "nt_min_mb=100,nt_mask=1" "nt_min_mb=100,nt_mask=3" "nt_min_mb=100,nt_mask=7" "nt_min_mb=100,nt_mask=15" "nt_min_mb=100,nt_mask=31" "nt_min_mb=50,nt_mask=31" "nt_min_mb=25,nt_mask=31" "nt_min_mb=200,nt_mask=31" "nt_min_mb=1,nt_mask=31" > gpurun_out/r4/exp4_knobs.txt 2>&1
python tools/gpu_layers.py fp16 32 256 small nt_min_mb=50 nt_mask=31 > gpurun_out/r4/exp4_layers_nt31.txt 2>&1
