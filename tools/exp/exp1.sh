set -e
mkdir -p gpurun_out/r4
python tools/gpu_kernel_sweep.py expand_dw fp16 32 256 small -- "" "irbx_grid=768" "irbx_grid=1536" "irbx_grid=512" "irbx_ablate=32" "irbx_ablate=33" "irbx_ablate=1" "irbx_ablate=4" "irbx_ablate=8" "" > gpurun_out/r4/exp1_sweep.txt 2>&1
for shape in "32 32 256 32 0" "64 64 128 32 0" "96 32 256 32 64"; do
  for abl in 0 1 32; do
    python tools/gpu_block.py $shape 10 1 0 1 $abl >> gpurun_out/r4/exp1_stamps.txt 2>&1
  done
done
