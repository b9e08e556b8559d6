#!/usr/bin/env python3
"""Quick perf + sanity loop for kernel work (GPU box): per-kernel-class device time of the BASELINE
config-2 workload from the engine's HIP-event hooks, plus a small-shape parity check."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import synth_input  # noqa: E402

M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
dev = torch.device("cuda:0")


def sanity():
    E = np.load(os.path.join(ROOT, "tests/golden/enhance_small64.npz"))
    spec = oracle.make_spec("small", 64)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    m = M.LowLightDiffusion(unet_variant="small", image_size=64)
    m.load_state_dict(sd); m = m.to(dev)
    low = synth_input("e2e64.low", (2, 3, 64, 64), -1.0, -0.4)
    torch.manual_seed(123)
    noise = torch.stack([torch.randn(2, 3, 64, 64) for _ in range(4)])
    for cd in ("fp32", "fp16"):
        m.compute_dtype = cd
        out = m.enhance(low.to(dev), 4, noise=noise, return_intermediate=True)
        e = (out.intermediate[-1].cpu().double() - torch.from_numpy(E["latents_3"]).double()).abs().max().item()
        print(f"sanity small@64 {cd}: final latent max-abs err {e:.3e}", flush=True)


def perf(dtype="fp16", B=32, size=256, variant="small", iters=5):
    m = M.LowLightDiffusion(unet_variant=variant, image_size=size, compute_dtype=dtype).to(dev)
    low = torch.rand(B, 3, size, size, device=dev) * 2 - 1
    for _ in range(2):
        m.enhance(low, 4)
    torch.cuda.synchronize()
    h = m.unet._prepare(B, dev)[0]
    t0 = time.perf_counter()
    for _ in range(iters):
        m.enhance(low, 4)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    h.profile_begin(N.K_DW | N.K_GEMM | N.K_CONV3 | N.K_SE)
    for _ in range(iters):
        m.enhance(low, 4)
    torch.cuda.synchronize()
    print(f"{variant}@{size} {dtype} B={B}: {wall*1e3:.2f} ms/enhance  {B/wall:.1f} img/s  ({wall*1e3/4:.2f} ms/forward)")
    for name, cls in [("dwconv", N.K_DW), ("pw_gemm", N.K_GEMM), ("conv3x3", N.K_CONV3), ("se", N.K_SE)]:
        ms, n, nb = h.profile_end(cls)
        print(f"   {name:8s} {ms/iters/4:7.3f} ms/forward  {n//iters//4:4d} launches  {nb/(ms*1e-3)/1e9 if ms else 0:8.1f} GB/s alg")
    sys.stdout.flush()


if __name__ == "__main__":
    sanity()
    perf("fp16", 32)
    if len(sys.argv) > 1 and sys.argv[1] == "all":
        perf("fp32", 8)
        perf("bf16", 32)
        perf("fp16", 8, 512, "large", 2)
