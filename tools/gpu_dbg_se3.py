#!/usr/bin/env python3
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from oracle.weightgen import synth_tensor
M = importlib.import_module("cv-diffusion-model_amd")
dev = torch.device("cuda:0")
N = importlib.import_module("cv-diffusion-model_amd._native")
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    N.check(N.lib().llie_tune(k.encode(), int(v)))
    print("knob", k, v)
spec = oracle.make_spec("small", 64)
sd_a = oracle.synth_state_dict(oracle.param_shapes(spec))
sd_b = {k: synth_tensor("other:" + k, tuple(v.shape)) for k, v in sd_a.items()}
low = (torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
noise = torch.stack(oracle.draw_noise(2, 64, 4, seed=2)).to(dev)
for cd in ("fp16",):
    m = M.LowLightDiffusion(unet_variant="small", image_size=64, compute_dtype=cd)
    m.load_state_dict(sd_a)
    m = m.to(dev).eval()
    ya = m.enhance(low, 4, noise=noise)
    ya0 = ya.clone()
    backup = {k: p.data.clone() for k, p in m.named_parameters()}
    for k, p in m.named_parameters():
        p.data.copy_(sd_b[k].to(dev))
    yb = m.enhance(low, 4, noise=noise)
    print(cd, "ya intact after yb:", torch.equal(ya, ya0), flush=True)
    fresh = M.LowLightDiffusion(unet_variant="small", image_size=64, compute_dtype=cd)
    fresh.load_state_dict(sd_b)
    yb_ref = fresh.to(dev).eval().enhance(low, 4, noise=noise)
    print(cd, "yb==yb_ref", torch.equal(yb, yb_ref), "ya intact:", torch.equal(ya, ya0), flush=True)
    for k, p in m.named_parameters():
        p.data.copy_(backup[k])
    yc = m.enhance(low, 4, noise=noise)
    torch.cuda.synchronize()
    print(cd, "yc==ya0", torch.equal(yc, ya0), "ya intact:", torch.equal(ya, ya0), "yc==yb", torch.equal(yc, yb), (yc - ya0).abs().max().item(), flush=True)
    yd = m.enhance(low, 4, noise=noise)
    print(cd, "yd==ya0", torch.equal(yd, ya0), "yd==yc", torch.equal(yd, yc), flush=True)
