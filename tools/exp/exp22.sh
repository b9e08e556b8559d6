set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_training.py -x -q -m gpu > gpurun_out/r4/exp22_tests.txt 2>&1 || { tail -30 gpurun_out/r4/exp22_tests.txt; exit 1; }
tail -4 gpurun_out/r4/exp22_tests.txt
timeout -k 10 120 python tools/gpu_train_perf.py 8 bf16 20 > gpurun_out/r4/exp22_train.txt 2>&1
timeout -k 10 120 python tools/gpu_train_perf.py 8 bf16 20 >> gpurun_out/r4/exp22_train.txt 2>&1
timeout -k 10 120 python tools/gpu_train_perf.py 32 bf16 10 >> gpurun_out/r4/exp22_train.txt 2>&1
