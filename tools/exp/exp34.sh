mkdir -p gpurun_out/r4
timeout -k 10 600 python tools/gpu_knobs.py 32 "" "enhance_split=3" "enhance_split=4" "enhance_split=1" "" "enhance_split=3" "enhance_split=4" > gpurun_out/r4/exp34.txt 2>&1; grep "B=" gpurun_out/r4/exp34.txt
