set -e
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/full_tests.txt 2>&1 || { tail -40 gpurun_out/r4/full_tests.txt; exit 1; }
tail -3 gpurun_out/r4/full_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -c 300
