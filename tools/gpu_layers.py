#!/usr/bin/env python3
"""Per-launch table of one UNet forward inside `enhance` (GPU box): every kernel launch of the last of 4 steps with its
operator tag, device time (HIP events on the launch stream) and algorithmic GB/s.  Usage: gpu_layers.py [dtype] [B] [size] [variant] [knob=value ...]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    variant = sys.argv[4] if len(sys.argv) > 4 else "small"
    dev = torch.device("cuda:0")
    for kv in sys.argv[5:]:  # engine knobs: name=value
        k, v = kv.split("=")
        N.check(N.lib().llie_tune(k.encode(), int(v)))
        print(f"# knob {k}={v}")
    m = M.LowLightDiffusion(unet_variant=variant, image_size=size, compute_dtype=dtype).to(dev)
    low = torch.rand(B, 3, size, size, device=dev) * 2 - 1
    for _ in range(2):
        m.enhance(low, 4)
    torch.cuda.synchronize()
    h = m.unet._prepare(B, dev)[0]
    reps = 3
    acc = None
    for _ in range(reps):
        h.profile_begin(31)
        m.enhance(low, 4)
        torch.cuda.synchronize()
        rows = h.profile_dump()
        per = len(rows) // 4
        rows = rows[3 * per:]  # last step
        if acc is None:
            acc = [list(r) for r in rows]
        else:
            for a, r in zip(acc, rows):
                a[3] += r[3]
    tot = 0.0
    print(f"# {variant}@{size} {dtype} B={B}: one forward, {len(acc)} launches (mean of {reps})")
    bytag = {}
    for cls, name, tag, ms, b in acc:
        ms /= reps
        tot += ms
        print(f"{tag:34s} {name[:58]:58s} {ms*1e3:9.1f} us {b/1e6:9.1f} MB {b/(ms*1e-3)/1e9 if ms else 0:8.0f} GB/s")
        bytag[tag] = bytag.get(tag, 0.0) + ms
    print(f"# total {tot:.3f} ms / forward")
    bykern = {}
    for cls, name, tag, ms, b in acc:
        k = (tag.split(" hid=")[0] if tag.startswith("irb") else tag) + " | " + name[:40]
        bykern[k] = bykern.get(k, 0.0) + ms / reps
    print("# per (operator, kernel):")
    for k, ms in bykern.items():
        print(f"#K  {k:80s} {ms*1e3:9.1f} us")
    print("# per operator:")
    for tag, ms in bytag.items():
        print(f"#   {tag:34s} {ms*1e3:9.1f} us")


if __name__ == "__main__":
    main()
