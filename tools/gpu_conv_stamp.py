#!/usr/bin/env python3
"""Up-sampling conv (conv.hip, MODE 1) alone on the GPU box: device time per launch and the in-kernel cycle stamps of the
diagnostic build (llie_tune("conv_stamp", 1)).  usage: gpu_conv_stamp.py [C Hi B reps]"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = importlib.import_module("cv-diffusion-model_amd._native")


def main():
    a = [int(v) for v in sys.argv[1:]]
    Cc, Hi, B, reps = (a + [256, 32, 32, 20][len(a):])[:4]
    dev = torch.device("cuda:0")
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    x = (torch.randn(B, Hi * Hi, Cc, device=dev)).half()
    w = (torch.randn(9, Cc, Cc, device=dev) / (3 * Cc ** 0.5)).half()
    bias = torch.zeros(Cc, device=dev)
    out = torch.empty(B, 4 * Hi * Hi, Cc, device=dev, dtype=torch.half)
    stats = torch.empty(B, int(L.llie_conv3x3_tiles(2 * Hi, 2 * Hi)), 2, Cc, device=dev)

    def run():
        N.check(L.llie_conv3x3(1, 1, x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), stats.data_ptr(), B, Hi, Hi, Cc, Cc, st))
    for stamp in (0, 1):
        N.check(L.llie_tune(b"conv_stamp", stamp))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"upconv C={Cc} {Hi}x{Hi} -> {2*Hi}x{2*Hi} B={B} stamp={stamp}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
    o = (C.c_double * 8)()
    N.check(L.llie_debug_conv_stamps(o))
    names = ["patch commit (blend + ds_write)", "first W tile staged + barrier", "operand ds_reads", "MFMAs", "next W tile staged / prefetched", "tap barrier", "epilogue"]
    tot = sum(o[:7])
    print(f"mean shader cycles per wave over {o[7]:.0f} waves: total {tot:.0f}\n  " + "\n  ".join(f"{n:34s} {v:9.0f} ({100 * v / tot:4.1f}%)" for n, v in zip(names, o)))
    N.check(L.llie_tune(b"conv_stamp", 0))


if __name__ == "__main__":
    main()
