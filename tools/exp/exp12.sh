set -e
mkdir -p gpurun_out/r4
python tools/gpu_knobs.py 32 "" "enhance_stagger=3" "enhance_stagger=6" "enhance_stagger=8" "enhance_stagger=10" "enhance_stagger=12" "enhance_stagger=15" "enhance_stagger=18" "enhance_stagger=22" "enhance_stagger=26" > gpurun_out/r4/exp12_stagger.txt 2>&1
