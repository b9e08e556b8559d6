set -e
mkdir -p gpurun_out/r4
python tools/gpu_kernel_sweep.py expand_dw fp16 32 256 small -- "" "irbx_var=1" "irbx_var=2" "irbx_var=3" "irbx_grid=512" "irbx_grid=512,irbx_var=1" "" > gpurun_out/r4/exp2_sweep.txt 2>&1
python -m pytest tests/test_gpu_round2.py -x -q -k "recompute or irbx or race" > gpurun_out/r4/exp2_tests.txt 2>&1 || true
for shape in "32 32 256 32 0" "64 64 128 32 0"; do
    python tools/gpu_block.py $shape 10 1 0 1 0 >> gpurun_out/r4/exp2_stamps.txt 2>&1
done
