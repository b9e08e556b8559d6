#!/usr/bin/env python3
"""Which engine knob breaks batch invariance?  Runs enhance on B=6 and on its two halves (small@64, 4 steps) per compute dtype
and knob setting and reports whether the concatenated halves equal the full batch bit for bit.
usage: gpu_shard_bisect.py ["knob=v,knob=v" ...]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
L = N.lib()
dev = torch.device("cuda:0")
RESET = {"irbx": 1, "pwx": 1, "irbx_dwv": 1, "enhance_split": 2, "ztot": 1}

for dtype in (None, "fp16", "bf16"):
    torch.manual_seed(3)
    m = M.LowLightDiffusion(unet_variant="small", image_size=64, compute_dtype=dtype).to(dev).eval()
    low = torch.rand(6, 3, 64, 64, device=dev) * 2 - 1
    noise = torch.randn(4, 6, 3, 64, 64, device=dev)
    for setting in sys.argv[1:] or [""]:
        knobs = dict(kv.split("=") for kv in setting.split(",") if kv)
        for k, v in knobs.items():
            N.check(L.llie_tune(k.encode(), int(v)))
        full = m.enhance(low, 4, noise=noise, return_noise_pred=True)
        parts = [m.enhance(low[a:b], 4, noise=noise[:, a:b], return_noise_pred=True) for a, b in ((0, 3), (3, 6))]
        same = torch.equal(torch.cat([p.enhanced for p in parts]), full.enhanced)
        eps = [bool(torch.equal(torch.cat([p.noise_pred[i] for p in parts]), full.noise_pred[i])) for i in range(4)]
        print(f"dtype={dtype} [{setting or 'defaults'}]: equal={same} per-step eps equal={eps}", flush=True)
        for k in knobs:
            N.check(L.llie_tune(k.encode(), RESET.get(k, 0)))
