#!/usr/bin/env python3
"""What a kernel boundary costs behind a write-heavy kernel, by store flavour (GPU box).
A chain of [streaming kernel writing / copying N MiB] -> [trivial dependent kernel] pairs in a captured graph against the same
streaming kernels alone: the difference per pair is the price of the extra boundary plus the drain of the dirty L2 lines."""
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
src = torch.empty(2 * GB, dtype=torch.uint8, device=dev).random_(0, 255)
dst = torch.empty(2 * GB, dtype=torch.uint8, device=dev)
flag = torch.zeros(64, dtype=torch.int32, device=dev)
NPAIR = 16


def chain(mib, r, w, mode, tiny):
    units = (mib << 20) // 16384 // max(r, w)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        def body():
            for i in range(NPAIR):
                off = (i % 4) * (mib << 20)
                N.check(L.llie_rw_probe(src.data_ptr() + off, dst.data_ptr() + off, units, r, w, mode, st.cuda_stream))
                if tiny:
                    N.check(L.llie_rw_probe(src.data_ptr(), flag.data_ptr(), 1, -1, 0, 0, st.cuda_stream))
        body()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            body()
        for _ in range(3):
            g.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            g.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / 10 / NPAIR * 1e3  # us per pair


for mib in (64, 256):
    for name, r, w in (("write", 0, 1), ("copy", 1, 1), ("read 1 : write 4", 1, 4)):
        for mode, mname in ((0, "plain"), (1, "nt"), (2, "sc1"), (3, "sc0 sc1")):
            a = chain(mib, r, w, mode, False)
            b = chain(mib, r, w, mode, True)
            print(f"{mib:4d} MiB {name:18s} {mname:8s} stores: kernel alone {a:7.1f} us, + trivial dependent kernel {b:7.1f} us  (boundary {b - a:5.1f} us)", flush=True)
