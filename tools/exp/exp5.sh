set -e
mkdir -p gpurun_out/r4
python tools/gpu_knobs.py 32 "" "nt_mask=5" "irbx_grid4=512" "irbx_grid4=256" "irbx_grid4=1024" "irbx_grid6=512" "irbx_grid6=1024" "irbx_grid6=256" "irbx_grid2=1536" "irbx_grid2=768" "nt_mask=0" > gpurun_out/r4/exp5_knobs.txt 2>&1
