set -e
mkdir -p gpurun_out/r4
python tools/gpu_kernel_sweep.py expand_dw fp16 32 256 small -- "" "irbx_var=4" "" "irbx_var=4" > gpurun_out/r4/exp17_sweep.txt 2>&1
python tools/gpu_knobs.py 32 "" "irbx_var=4" > gpurun_out/r4/exp17_step.txt 2>&1
