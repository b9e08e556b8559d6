#!/usr/bin/env python3
"""One InvertedResidualBlock shape in a loop (GPU box): the workload for rocprofv3 passes over a single kernel family, and
the reader of the diagnostic in-kernel cycle stamps of expand_dw (llie_tune("irbx_stamp", 1)).
usage: gpu_block.py [cin cout hw batch split reps irbx dbuf stamp]"""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")


def main():
    a = [int(v) for v in sys.argv[1:]]
    cin, cout, hw, b, split, reps, irbx, dbuf, stamp, ablate = (a + [32, 32, 256, 32, 0, 10, 1, 1, 0, 0][len(a):])[:10]
    dev = torch.device("cuda:0")
    L = N.lib()
    N.check(L.llie_tune(b"irbx", irbx))
    N.check(L.llie_tune(b"irbx_dbuf", dbuf))
    N.check(L.llie_tune(b"irbx_stamp", stamp))
    N.check(L.llie_tune(b"irbx_ablate", ablate))
    blk = M.InvertedResidualBlock(cin, cout, 128, concat_split=split).to(dev)
    blk.compute_dtype = "fp16"
    x = torch.rand(b, cin, hw, hw, device=dev) * 4 - 2
    te = torch.rand(b, 128, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(2):
            blk(x, te)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            blk(x, te)
        torch.cuda.synchronize()
    print(f"irb {cin}->{cout} {hw}x{hw} B={b} irbx={irbx} dbuf={dbuf} ablate={ablate}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per block "
          f"(incl. NCHW<->NHWC conversion of the operator boundary)")
    if stamp:
        out = (C.c_double * 10)()
        N.check(L.llie_debug_irbx_stamps(out))
        names = ["x wait (vmcnt)", "activate + ds_write", "next-tile loads + halo flags", "tile-top barrier", "pool flush", "chunk-top barrier + flush",
                 "expand MFMAs + epilogue", "barrier behind them", "depthwise phase + stores"]
        tot = sum(out[:9])
        print(f"expand_dw mean shader cycles per wave over {out[9]:.0f} waves (s_memtime): total {tot:.0f}\n  "
              + "\n  ".join(f"{n:30s} {v:9.0f} ({100 * v / tot:4.1f}%)" for n, v in zip(names, out)))


if __name__ == "__main__":
    main()
