#!/bin/bash
# A/B timing of two builds of the library on ONE box: ab/libA.so and ab/libB.so alternate as cv-diffusion-model_amd/libllie_hip.so
# (box-to-box variance is +-1.5 %, more than most kernel changes).  usage: tools/ab_libs.sh [B] [rounds]   (inside a gpurun call)
B=${1:-32}; R=${2:-2}
for r in $(seq $R); do
  for v in A B; do
    cp ab/lib$v.so cv-diffusion-model_amd/libllie_hip.so
    echo "== lib$v"; timeout -k 10 120 python tools/gpu_knobs.py $B "" 2>&1 | grep "ms/step" | tail -1
  done
done
