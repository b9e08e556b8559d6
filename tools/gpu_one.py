#!/usr/bin/env python3
"""Runs ONE kernel shape a few times (for rocprofv3 --pmc passes).  usage: gpu_one.py {gemm:<idx>|dw:<H>:<C>} [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune as T  # noqa: E402
import torch  # noqa: E402

what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = 32
if what.startswith("gemm:"):
    name, kind, P, segs, n = T.SHAPES[int(what.split(":")[1])]
    us, gbs, tf = T.gemm(kind, B * P, segs, n, P)
    print(name, f"{us:.1f} us {gbs:.0f} GB/s {tf:.0f} TF")
else:
    _, H, Cc = what.split(":")
    us, gbs = T.dw(B, int(H), int(Cc))
    print("dw", H, Cc, f"{us:.1f} us {gbs:.0f} GB/s")
torch.cuda.synchronize()
