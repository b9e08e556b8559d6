set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3_entry or downsample_upsample or unpinned" > gpurun_out/r4/exp30_t0.txt 2>&1 || { tail -40 gpurun_out/r4/exp30_t0.txt; exit 1; }
tail -3 gpurun_out/r4/exp30_t0.txt
timeout -k 10 300 python tools/gpu_kernel_sweep.py conv -- "" "" > gpurun_out/r4/exp30_sweep.txt 2>&1
grep -E "^==|mode=1" gpurun_out/r4/exp30_sweep.txt
