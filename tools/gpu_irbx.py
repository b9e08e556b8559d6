#!/usr/bin/env python3
"""irbx.hip (statistics-only expand + tile-fused expand/depthwise) on the GPU box: operator-level agreement with the
unfused path and the CPU oracle (fp16 / bf16), whole-network agreement, and an A/B timing of one forward."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from oracle import unet_ref  # noqa: E402
from oracle.weightgen import synth_tensor  # noqa: E402
from conftest import synth_input  # noqa: E402

M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
dev = torch.device("cuda:0")
L = N.lib()


def ops():
    worst = 0.0
    for cd in ("fp16", "bf16"):
        for cin, cout, hw, b, split in [(32, 32, 16, 2, 0), (32, 32, 64, 3, 0), (32, 64, 32, 2, 0), (64, 64, 32, 2, 0), (64, 128, 16, 1, 0),
                                        (96, 32, 64, 2, 64), (96, 32, 16, 2, 64), (32, 32, 256, 1, 0)]:
            name = f"xi_{cin}_{cout}_{hw}"
            blk = M.InvertedResidualBlock(cin, cout, 128, concat_split=split)
            blk.load_state_dict({k: synth_tensor(name + "." + k, tuple(v.shape)) for k, v in blk.state_dict().items()})
            blk = blk.to(dev)
            blk.compute_dtype = cd
            sd = {name + "." + k: v.detach().cpu() for k, v in blk.state_dict().items()}
            x = synth_input(name + ".x", (b, cin, hw, hw), -2, 2)
            te = synth_input(name + ".temb", (b, 128), -1, 1)
            ref = unet_ref.irb_forward(sd, name, x, te)
            ys = []
            for v in (0, 1):
                N.check(L.llie_tune(b"irbx", v))
                ys.append(blk(x.to(dev), te.to(dev)).cpu())
            N.check(L.llie_tune(b"irbx", 1))
            sc = max(1.0, ref.abs().max().item())
            e0 = (ys[0] - ref).abs().max().item() / sc
            e1 = (ys[1] - ref).abs().max().item() / sc
            d = (ys[0] - ys[1]).abs().max().item() / sc
            r0 = ((ys[0] - ref).norm() / ref.norm()).item()
            r1 = ((ys[1] - ref).norm() / ref.norm()).item()
            print(f"{cd} irb {cin}->{cout} {hw}x{hw} B={b} split={split}: max-abs/scale unfused {e0:.2e} fused {e1:.2e} "
                  f"(fused vs unfused {d:.2e}); rel L2 unfused {r0:.2e} fused {r1:.2e}", flush=True)
            worst = max(worst, r1 / max(r0, 1e-9))
    print(f"worst fused/unfused relative-L2 ratio: {worst:.2f}")


def net(cd="fp16", B=2):
    m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype=cd).to(dev)
    low = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(dev)
    noise = torch.randn(4, B, 3, 256, 256, generator=torch.Generator().manual_seed(5)).to(dev)
    outs = []
    for v in (0, 1, 1):
        N.check(L.llie_tune(b"irbx", v))
        o = m.enhance(low, 4, noise=noise, return_intermediate=True, return_noise_pred=True)
        outs.append((o.noise_pred[0].clone(), o.intermediate[-1].clone()))
    rel = (outs[0][0] - outs[1][0]).abs().max().item() / outs[0][0].abs().max().item()
    print(f"net small@256 {cd} B={B}: first noise_pred fused vs unfused rel max {rel:.2e}; "
          f"final latents max diff {(outs[0][1] - outs[1][1]).abs().max().item():.3e} (absmax {outs[0][1].abs().max().item():.1f}); "
          f"fused run twice bit-equal: {torch.equal(outs[1][1], outs[2][1])}", flush=True)
    # sub-batch invariance of the fused path
    one = m.enhance(low[1:2], 4, noise=noise[:, 1:2], return_intermediate=True).intermediate[-1]
    print(f"   fused: row 1 of B={B} == B=1 run bitwise: {torch.equal(outs[1][1][1:2], one)}", flush=True)


def perf(B=32, cd="fp16"):
    m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype=cd).to(dev)
    low = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
    for cfg in [("irbx", 0, {}), ("irbx", 1, {"irbx_dbuf": 1, "irbx_tiles": 4}), ("irbx", 1, {"irbx_dbuf": 0, "irbx_tiles": 4}),
                ("irbx", 0, {}), ("irbx", 1, {"irbx_dbuf": 1, "irbx_tiles": 4})]:
        N.check(L.llie_tune(cfg[0].encode(), cfg[1]))
        for k, v in cfg[2].items():
            N.check(L.llie_tune(k.encode(), v))
        for _ in range(3):
            m.enhance(low, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.enhance(low, 4)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5
        print(f"{cfg}: {wall*1e3:.2f} ms/enhance = {B/wall:.1f} img/s", flush=True)
    # per-launch table of the fused kernels
    h = m.unet._prepare(B, dev)[0]
    h.profile_begin(31)
    m.enhance(low, 4)
    torch.cuda.synchronize()
    rows = h.profile_dump()
    per = len(rows) // 4
    tot = 0.0
    for cls, name, tag, ms, b in rows[3 * per:]:
        tot += ms
        if "expand" in name or "irb P=65536" in tag or "hid=256" in tag:
            print(f"{tag:34s} {name[:50]:50s} {ms*1e3:9.1f} us {b/1e6:9.1f} MB {b/(ms*1e-3)/1e9 if ms else 0:8.0f} GB/s")
    print(f"# total {tot:.3f} ms / forward (eager, events)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("ops", "all"):
        ops()
    if what in ("net", "all"):
        net("fp16")
        net("bf16")
    if what in ("perf", "all"):
        perf()
