mkdir -p gpurun_out/r4
timeout -k 10 600 python tools/gpu_knobs.py 32 "" "skip_small=4" "skip_small=8" "skip_small=12" "" "skip_small=4" "skip_small=8" "skip_small=12" > gpurun_out/r4/exp35.txt 2>&1; grep "B=" gpurun_out/r4/exp35.txt
timeout -k 10 600 python tools/gpu_knobs.py 1 "" "skip_small=4" "skip_small=8" "skip_small=12" "" > gpurun_out/r4/exp35b.txt 2>&1; grep "B=" gpurun_out/r4/exp35b.txt
