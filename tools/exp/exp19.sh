set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "se_mlp" > gpurun_out/r4/exp19_tests.txt 2>&1 || true
tail -15 gpurun_out/r4/exp19_tests.txt
