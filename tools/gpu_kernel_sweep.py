#!/usr/bin/env python3
"""A/B of engine knobs on ONE kernel family inside one process (GPU box): for every knob setting, the per-launch device
time (HIP events, eager single chain, last of 4 steps, mean of `reps` calls) of the launches whose kernel name contains
`pattern`, listed per operator tag, plus the forward total.
usage: gpu_kernel_sweep.py pattern [dtype B size variant] -- "knob=v,knob=v" "knob=v" ...   ("" = defaults)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")


def main():
    sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
    head, settings = sys.argv[1:sep], sys.argv[sep + 1:] or [""]
    pattern = head[0]
    dtype, B, size, variant = (head[1:] + ["fp16", "32", "256", "small"][len(head) - 1:])[:4]
    B, size = int(B), int(size)
    dev = torch.device("cuda:0")
    L = N.lib()
    m = M.LowLightDiffusion(unet_variant=variant, image_size=size, compute_dtype=dtype).to(dev)
    low = torch.rand(B, 3, size, size, device=dev) * 2 - 1
    reps = 3
    defaults = {}
    for st in settings:
        knobs = dict(kv.split("=") for kv in st.split(",") if kv)
        for k in defaults:
            if k not in knobs:
                N.check(L.llie_tune(k.encode(), defaults[k]))
        for k, v in knobs.items():
            defaults.setdefault(k, 0)
            N.check(L.llie_tune(k.encode(), int(v)))
        for _ in range(2):
            m.enhance(low, 4)
        torch.cuda.synchronize()
        h = m.unet._prepare(B, dev)[0]
        acc = {}
        tot = 0.0
        for _ in range(reps):
            h.profile_begin(31)
            m.enhance(low, 4)
            torch.cuda.synchronize()
            rows = h.profile_dump()
            per = len(rows) // 4
            for cls, name, tag, ms, b in rows[3 * per:]:
                tot += ms / reps
                if pattern in name:
                    key = (tag.split(" hid=")[0], name[:48])
                    e = acc.setdefault(key, [0.0, 0, b])
                    e[0] += ms / reps
                    e[1] += 1
        print(f"== [{st or 'defaults'}]  forward {tot:.3f} ms; '{pattern}' {sum(e[0] for e in acc.values()) * 1e3:.1f} us", flush=True)
        for (tag, name), (ms, n, b) in acc.items():
            n //= reps
            print(f"   {tag:28s} {name:48s} x{n}  {ms / n * 1e3:8.1f} us each  {b / (ms / n * 1e-3) / 1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
