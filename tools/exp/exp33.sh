mkdir -p gpurun_out/r4
timeout -k 10 400 python tools/gpu_knobs.py 32 "" "skip_small=1" "skip_small=2" "skip_small=3" "" "skip_small=3" > gpurun_out/r4/exp33_b32.txt 2>&1; grep "B=" gpurun_out/r4/exp33_b32.txt
timeout -k 10 400 python tools/gpu_knobs.py 1 "" "skip_small=1" "skip_small=2" "skip_small=3" "" > gpurun_out/r4/exp33_b1.txt 2>&1; grep "B=" gpurun_out/r4/exp33_b1.txt
