#!/usr/bin/env python3
"""Race screen for the recompute kernels: the same InvertedResidualBlock launched on two streams at once (different inputs,
own handles and workspaces), many rounds; every output must equal the block's solo result bit for bit.  Uneven load
(the other stream runs a different shape) is what makes latent LDS / visibility races show."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
dev = torch.device("cuda:0")
L = N.lib()


def mk(cin, cout, hw, b, split, seed):
    torch.manual_seed(seed)
    blk = M.InvertedResidualBlock(cin, cout, 128, concat_split=split).to(dev)
    blk.compute_dtype = "fp16"
    x = torch.rand(b, cin, hw, hw, device=dev) * 4 - 2
    te = torch.rand(b, 128, device=dev) * 2 - 1
    return blk, x, te


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    cfgs = [(32, 32, 256, 16, 0), (96, 32, 256, 8, 64), (64, 64, 128, 16, 0), (32, 32, 128, 16, 0)]
    for knobs in [{"irbx": 1, "irbx_dbuf": 1}, {"irbx": 1, "irbx_dbuf": 0}, {"irbx": 0}]:
        for k, v in knobs.items():
            N.check(L.llie_tune(k.encode(), v))
        items = [mk(*c, seed=i) for i, c in enumerate(cfgs)]
        with torch.no_grad():
            solo = []
            for blk, x, te in items:
                y = blk(x, te)
                y2 = blk(x, te)
                assert torch.equal(y, y2), "solo run not reproducible"
                solo.append(y.clone())
            torch.cuda.synchronize()
            streams = [torch.cuda.Stream() for _ in items]
            bad = [0] * len(items)
            for r in range(rounds):
                outs = []
                for (blk, x, te), st in zip(items, streams):
                    with torch.cuda.stream(st):
                        outs.append(blk(x, te))
                torch.cuda.synchronize()
                for i, (o, s) in enumerate(zip(outs, solo)):
                    if not torch.equal(o, s):
                        bad[i] += 1
                        if bad[i] == 1:
                            d = (o != s)
                            rows = d.flatten(1).any(1).nonzero().flatten().tolist()
                            ch = d.any(0).flatten(1).any(1).nonzero().flatten().tolist()
                            print(f"   cfg {cfgs[i]} round {r}: rows {rows[:8]} channels {ch[:16]} ({len(ch)} ch) n={int(d.sum())} max {(o - s).abs().max().item():.3e}", flush=True)
        print(f"{knobs}: mismatching rounds per config {bad} of {rounds}", flush=True)


if __name__ == "__main__":
    main()
