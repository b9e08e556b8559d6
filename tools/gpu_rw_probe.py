#!/usr/bin/env python3
"""HBM streaming rates at different read : write mixes (GPU box): what the write-dominated launches can be held to."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = importlib.import_module("cv-diffusion-model_amd._native")
L = N.lib()
dev = torch.device("cuda:0")
GB = 1 << 30
src = torch.empty(4 * GB, dtype=torch.uint8, device=dev).random_(0, 255)
dst = torch.empty(4 * GB, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for nt in (0, 1):
    for r, w in [(1, 0), (0, 1), (1, 1), (1, 2), (1, 4), (2, 1), (4, 1)]:
        units = (4 * GB // 16384) // max(r, w)
        for _ in range(2):
            N.check(L.llie_rw_probe(src.data_ptr(), dst.data_ptr(), units, r, w, nt, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            N.check(L.llie_rw_probe(src.data_ptr(), dst.data_ptr(), units, r, w, nt, st))
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        rb, wb = units * r * 16384, units * w * 16384
        print(f"nt={nt} read:write {r}:{w}  {rb / GB:5.2f} GiB read {wb / GB:5.2f} GiB written  {t * 1e3:7.3f} ms  "
              f"read {rb / t / 1e12:5.2f} TB/s  write {wb / t / 1e12:5.2f} TB/s  total {(rb + wb) / t / 1e12:5.2f} TB/s", flush=True)
