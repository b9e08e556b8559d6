#!/usr/bin/env python3
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
dev = torch.device("cuda:0")
for cd in (None, "fp16"):
    for size, B in ((64, 2), (256, 4)):
        m = M.LowLightDiffusion(unet_variant="small", image_size=size, compute_dtype=cd).to(dev).eval()
        g = torch.Generator().manual_seed(1)
        low = (torch.rand(B, 3, size, size, generator=g) * 2 - 1).to(dev)
        nz = torch.randn(4, B, 3, size, size, generator=g).to(dev)
        outs = [m.enhance(low, 4, noise=nz).clone() for _ in range(5)]
        print(cd, size, B, [torch.equal(outs[0], o) for o in outs[1:]], [(outs[0] - o).abs().max().item() for o in outs[1:]], flush=True)
