#!/usr/bin/env python3
"""tests/golden/large_kat.npz: the `large` variant (BASELINE config 4's network) run by the REFERENCE in the build
container -- a 4-step `enhance` at image_size 64 (B=1: all noise predictions and pre-clamp latents) and one UNet forward
at image_size 128.  Same loader, weight generator and input recipe as tools/make_golden.py (nothing of the reference
is copied; the fixture holds its outputs only).  Runs only where /root/reference exists."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import OUT, fill_, load_ref_models_package, load_ref_unet_module, np32, synth_input  # noqa: E402


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    U = load_ref_unet_module()
    M = load_ref_models_package()
    g = {}
    model = M.LowLightDiffusion(unet_variant="large", image_size=64, num_inference_steps=4).eval()
    fill_(model)
    low = synth_input("e2eL64.low", (1, 3, 64, 64), -1.0, -0.4)
    preds = []
    h = model.unet.register_forward_hook(lambda mod, i, o: preds.append(o.clone()))
    torch.manual_seed(321)
    out = model.enhance(low, num_inference_steps=4, return_intermediate=True)
    h.remove()
    g["enhanced"] = np32(out.enhanced)
    for i, (p, z) in enumerate(zip(preds, out.intermediate)):
        g[f"noise_pred_{i}"] = np32(p)
        g[f"latents_{i}"] = np32(z)
    g["seed"] = np.array([321])
    m = U.create_efficient_unet("large", image_size=128, in_channels=6).eval()
    fill_(m, "unet.")
    x = synth_input("large128.x", (1, 6, 128, 128), -1.5, 1.5)
    t = torch.tensor([499], dtype=torch.long)
    g["unet128"] = np32(m(x, t))
    g["unet128_t"] = t.numpy()
    path = os.path.join(OUT, "large_kat.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
