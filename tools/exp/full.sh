set -e
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/full_tests.txt 2>&1 || { tail -40 gpurun_out/r4/full_tests.txt; exit 1; }
tail -3 gpurun_out/r4/full_tests.txt
