set -e
mkdir -p gpurun_out/r4
timeout -k 10 200 python tools/gpu_kernel_sweep.py pw_gemm -- "" "gemm_bk=32" "" > gpurun_out/r4/exp23_sweep.txt 2>&1
grep -E "^==|32->64|96->32" gpurun_out/r4/exp23_sweep.txt | head -40
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -m gpu > gpurun_out/r4/exp23_tests.txt 2>&1 || { tail -30 gpurun_out/r4/exp23_tests.txt; exit 1; }
tail -3 gpurun_out/r4/exp23_tests.txt
