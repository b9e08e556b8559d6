set -e
mkdir -p gpurun_out/r4
python tools/gpu_knobs.py 32 "" "gn_inline=0" > gpurun_out/r4/exp9_b32.txt 2>&1
python tools/gpu_knobs.py 1 "" "gn_inline=0" > gpurun_out/r4/exp9_b1.txt 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r4/exp9_tests.txt 2>&1 || true
tail -5 gpurun_out/r4/exp9_tests.txt
