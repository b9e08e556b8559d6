set -e
mkdir -p gpurun_out/r4
python tools/gpu_layers.py fp16 32 256 small > gpurun_out/r4/exp3_layers_def.txt 2>&1
python tools/gpu_layers.py fp16 32 256 small irbx_var=2 > gpurun_out/r4/exp3_layers_nt.txt 2>&1
python tools/gpu_kernel_sweep.py expand_dw fp16 32 256 small -- "irbx_var=2" "irbx_var=2,irbx_grid=512" "irbx_var=2,irbx_grid=1024" "irbx_var=2,irbx_grid=768" "irbx_var=2,irbx_grid=1536" "irbx_var=2,irbx_grid=2304" "irbx_var=2,irbx_grid=3072" "irbx_grid=1024" "irbx_grid=256"  > gpurun_out/r4/exp3_sweep.txt 2>&1
