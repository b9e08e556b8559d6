#!/usr/bin/env python3
"""Golden vectors of one training step of the reference (low_light_diffusion.py:140-171,250-277 + autograd):
small built at image_size=64 with the hash-generated weights, B=2, explicit timesteps / noise, MSE loss.
Stores the loss, the L2 norm of every parameter gradient (381 values) and a few small gradient tensors in
full.  Runs only where /root/reference exists; writes tests/golden/train_small64.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as G  # noqa: E402  (loader of the reference package + weight fill)

FULL = ["unet.final_conv.weight", "unet.final_conv.bias", "unet.init_conv.weight", "unet.time_mlp.3.bias",
        "unet.time_mlp.1.weight", "unet.final_norm.weight", "unet.mid_attn.norm.weight", "unet.mid_attn.to_out.1.bias",
        "unet.encoder_blocks.0.0.depthwise.weight", "unet.encoder_blocks.0.0.norm2.bias", "unet.decoder_blocks.3.0.skip.weight",
        "unet.encoder_blocks.1.0.se.fc1.bias", "unet.downsamplers.0.down.bias", "unet.upsamplers.2.conv.bias",
        "unet.decoder_blocks.3.2.time_mlp.1.bias", "unet.encoder_blocks.0.1.expand.weight"]


def main():
    M = G.load_ref_models_package()
    model = M.LowLightDiffusion(unet_variant="small", image_size=64, num_inference_steps=4).train()
    G.fill_(model)
    low = G.synth_input("train64.low", (2, 3, 64, 64), -1.0, -0.4)
    normal = G.synth_input("train64.normal", (2, 3, 64, 64), -1, 1)
    noise = G.synth_input("train64.noise", (2, 3, 64, 64), -2, 2)
    t = torch.tensor([500, 37])
    out = model(low, normal, timesteps=t, noise=noise)
    loss = torch.nn.functional.mse_loss(out["noise_pred"], out["noise"])
    loss.backward()
    res = {"loss": np.array(loss.item()), "timesteps": t.numpy()}
    keys, norms = [], []
    for k, p in model.named_parameters():
        keys.append(k)
        norms.append(p.grad.double().norm().item())
    res["keys"] = np.array(keys)
    res["grad_norms"] = np.array(norms)
    for k in FULL:
        res["grad:" + k] = dict(model.named_parameters())[k].grad.numpy().astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "train_small64.npz")
    np.savez_compressed(path, **res)
    print(path, os.path.getsize(path), "loss", loss.item(), len(keys), "params")


if __name__ == "__main__":
    main()
