#!/usr/bin/env python3
"""Step time of the shipping path (hipGraph replay of `enhance`, small@256, 4 steps) under engine knob settings.
usage: gpu_knobs.py [B] "knob=v,knob=v" "..."   (an empty string = defaults).  Timing ablation knobs give garbage results."""
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
args = sys.argv[1:]
B = int(args.pop(0)) if args and args[0].isdigit() else 32
m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype="fp16").to(dev).eval()
low = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
defaults = {}


def run(setting, steps=10):
    knobs = dict(kv.split("=") for kv in setting.split(",") if kv)
    for k, v in knobs.items():
        defaults.setdefault(k, 0)
        N.check(L.llie_tune(k.encode(), int(v)))
    for _ in range(3):
        m.enhance(low, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        m.enhance(low, 4)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    for k in knobs:
        N.check(L.llie_tune(k.encode(), {"irbx": 1, "irbx_dbuf": 0, "irbx_mask": 7, "irbx_tiles": 4, "gemm_bk128": 1024, "enhance_split": 2, "ztot": 1, "pwx": 1, "irbx_dwv": 1, "gram": 1}.get(k, 0)))
    return ms


for rep in range(2):
    for setting in (args or [""]):
        ms = run(setting)
        print(f"B={B} [{setting or 'defaults'}]: {ms:.3f} ms/step  {B / ms * 1e3:.1f} img/s", flush=True)
