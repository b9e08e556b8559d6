#!/usr/bin/env python3
"""A few graph-replayed `enhance` calls (small@256, fp16) for `rocprofv3 --kernel-trace`: the workload of
tools/timeline_summary.py, which measures what the kernel boundaries of the replayed graph cost.
usage: gpu_timeline.py [B] [split] [steps]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
a = [int(v) for v in sys.argv[1:]]
B, split, steps = (a + [32, 1, 4][len(a):])[:3]
N.check(N.lib().llie_tune(b"enhance_split", split))
dev = torch.device("cuda:0")
m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype="fp16").to(dev).eval()
low = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
for _ in range(steps):
    m.enhance(low, 4)
torch.cuda.synchronize()
print("done")
