#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats -- python tools/gpu_trace_run.py B [knob=value ...]`: 24 replays of the
captured `enhance` graph (small@256, 4 steps, fp16) at batch B under engine knobs -- kernel durations as they are INSIDE the
graph (tools/gpu_layers.py times eager launches, which at B = 1 are several times longer than in the graph)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    N.check(N.lib().llie_tune(k.encode(), int(v)))
dev = torch.device("cuda:0")
m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype="fp16").to(dev).eval()
low = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
for _ in range(24):
    m.enhance(low, 4)
torch.cuda.synchronize()
