mkdir -p gpurun_out/r4
timeout -k 10 600 python tools/gpu_knobs.py 32 "" "skip_small=16" "skip_small=32" "skip_small=48" "" "skip_small=16" "skip_small=32" "skip_small=48" > gpurun_out/r4/exp39.txt 2>&1; grep "B=" gpurun_out/r4/exp39.txt
