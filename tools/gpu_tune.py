#!/usr/bin/env python3
"""Times individual pointwise-GEMM / depthwise launches at the shapes of small@256 B=32 (GPU box)."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = importlib.import_module("cv-diffusion-model_amd._native")
L = N.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream


def time_it(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def gemm(kind, M, segs, Nout, P, dtype=1, B=32, act=1):
    """kind 'k1': affine+relu6 prologue on all segs, stats; 'k3': gate on seg0, identity others, residual if 1 seg."""
    tdt = torch.float16 if dtype == 1 else torch.float32
    K = sum(segs)
    a = [torch.randn(M, c, device=dev, dtype=tdt) for c in segs]
    w = torch.randn(Nout, K, device=dev, dtype=tdt) * 0.05
    out = torch.empty(M, Nout, device=dev, dtype=tdt)
    sc = torch.rand(B, K, device=dev) + 0.5
    bi = torch.randn(B, K, device=dev) * 0.1
    tiles = P // L.llie_pw_gemm_tile_rows(P)
    stats = torch.empty(B * tiles * 2 * Nout, device=dev)
    res = torch.randn(M, Nout, device=dev, dtype=tdt) if (kind == "k3" and len(segs) == 1) else None
    arr = (N.GemmSeg * len(segs))()
    off = 0
    for i, c in enumerate(segs):
        if kind == "k1":
            arr[i] = N.GemmSeg(a[i].data_ptr(), c, sc.data_ptr() + off * 4, bi.data_ptr() + off * 4, K, act)
        elif i == 0:
            arr[i] = N.GemmSeg(a[i].data_ptr(), c, sc.data_ptr(), None, K, 0)
        else:
            arr[i] = N.GemmSeg(a[i].data_ptr(), c, None, None, 0, 0)
        off += c

    def run():
        N.check(L.llie_pw_gemm(dtype, arr, len(segs), w.data_ptr(), None, res.data_ptr() if res is not None else None,
                               out.data_ptr(), stats.data_ptr(), M, Nout, P, st()))
    us = time_it(run)
    es = 2 if dtype else 4
    nbytes = (M * K + M * Nout * (2 if res is not None else 1)) * es
    flops = 2.0 * M * K * Nout
    return us, nbytes / us / 1e3, flops / us / 1e6


def expand(M, segs, Nout, P, dtype=1, B=32):
    """The activation-stationary expand GEMM (llie_pw_expand) and the tile kernel on the same operands."""
    tdt = torch.float16 if dtype == 1 else torch.bfloat16
    K = sum(segs)
    a = [torch.randn(M, c, device=dev, dtype=tdt) for c in segs]
    w32 = torch.randn(Nout, K, device=dev) * 0.05
    wt = w32.to(tdt)
    wpack = torch.empty(Nout * K, device=dev, dtype=tdt)
    out = torch.empty(M, Nout, device=dev, dtype=tdt)
    sc = (torch.rand(B, K, device=dev) + 0.5) / 6
    bi = torch.randn(B, K, device=dev) * 0.1
    stats = torch.empty(B * (P // 128) * 2 * Nout, device=dev)
    arr = (N.GemmSeg * len(segs))()
    off = 0
    for i, c in enumerate(segs):
        arr[i] = N.GemmSeg(a[i].data_ptr(), c, sc.data_ptr() + off * 4, bi.data_ptr() + off * 4, K, 3)
        off += c
    N.check(L.llie_pw_expand(dtype, arr, len(segs), w32.data_ptr(), wpack.data_ptr(), out.data_ptr(), stats.data_ptr(), M, Nout, P, st()))
    L.llie_tune(b"pwx", 1)

    def run_new():  # the pack is a load-time cost in the engine: time the GEMM alone through the engine's launcher
        N.check(L.llie_pw_expand(dtype, arr, len(segs), None, wpack.data_ptr(), out.data_ptr(), stats.data_ptr(), M, Nout, P, st()))

    def run_old():
        N.check(L.llie_pw_gemm(dtype, arr, len(segs), wt.data_ptr(), None, None, out.data_ptr(), stats.data_ptr(), M, Nout, P, st()))
    nbytes = (M * K + M * Nout) * 2
    flops = 2.0 * M * K * Nout
    return time_it(run_new), time_it(run_old), nbytes, flops


def dw(B, H, C, dtype=1):
    tdt = torch.float16 if dtype == 1 else torch.float32
    x = torch.randn(B, H, H, C, device=dev, dtype=tdt)
    y = torch.empty_like(x)
    sc = torch.rand(B, C, device=dev) + 0.5
    bi = torch.randn(B, C, device=dev) * 0.1
    w = torch.randn(9, C, device=dev) * 0.3
    pool = torch.empty(B * L.llie_dwconv3x3_tiles(H, H) * C, device=dev)

    def run():
        N.check(L.llie_dwconv3x3(dtype, x.data_ptr(), y.data_ptr(), sc.data_ptr(), bi.data_ptr(), w.data_ptr(),
                                 pool.data_ptr(), B, H, H, C, st()))
    us = time_it(run)
    return us, 2 * x.numel() * x.element_size() / us / 1e3


SHAPES = [  # (name, kind, P, segs, N)   small@256, B=32
    ("enc0 K1 32->128", "k1", 65536, [32], 128), ("enc0 K3 128->32", "k3", 65536, [128], 32),
    ("dec3.0 K1 96->384", "k1", 65536, [64, 32], 384), ("dec3.0 K3 384+96->32", "k3", 65536, [384, 64, 32], 32),
    ("enc1.1 K1 64->256", "k1", 16384, [64], 256), ("enc1.1 K3 256->64", "k3", 16384, [256], 64),
    ("dec2.0 K1 192->768", "k1", 16384, [128, 64], 768), ("dec2.0 K3 768+192->64", "k3", 16384, [768, 128, 64], 64),
    ("enc2.1 K1 128->512", "k1", 4096, [128], 512), ("enc2.1 K3 512->128", "k3", 4096, [512], 128),
    ("dec1.0 K1 384->1536", "k1", 4096, [256, 128], 1536), ("dec1.0 K3 1536+384->128", "k3", 4096, [1536, 256, 128], 128),
    ("mid K1 256->1024", "k1", 1024, [256], 1024), ("mid K3 1024->256", "k3", 1024, [1024], 256),
    ("dec0.0 K1 512->2048", "k1", 1024, [256, 256], 2048), ("dec0.0 K3 2048+512->256", "k3", 1024, [2048, 256, 256], 256),
]

if __name__ == "__main__":
    B = 32
    if "gemm" in sys.argv[1:]:
        knobs = [("relu6 fp32 prologue", 1), ("clamp01(z/6) prologue", 3), ("relu6 fp32 prologue", 1), ("clamp01(z/6) prologue", 3)]
        print(f"{'shape':28s} " + " ".join(f"{k:>26s}" for k, _ in knobs))
        tot = {}
        for name, kind, P, segs, n in SHAPES:
            row = []
            for k, v in knobs:
                us, gbs, tf = gemm(kind, B * P, segs, n, P, act=v)
                tot[k] = tot.get(k, 0.0) + us
                row.append(f"{us:8.1f}us {tf:4.0f}TF {gbs:5.0f}GB/s")
            print(f"{name:28s} " + " ".join(f"{r:>26s}" for r in row), flush=True)
        print("sum(us):", {k: round(v, 1) for k, v in tot.items()})
    if "expand" in sys.argv[1:]:  # activation-stationary expand GEMM vs the tile kernel; usage: gpu_tune.py expand [B]
        i = sys.argv.index("expand")
        Bx = int(sys.argv[i + 1]) if len(sys.argv) > i + 1 else 32
        tot_n = tot_o = 0.0
        for rep in range(2):
            for name, kind, P, segs, n in SHAPES + [("enc3.0 K1 128->512", "k1", 1024, [128], 512)]:
                if kind != "k1" or sum(segs) < 128:
                    continue
                L.llie_tune(b"pwx_ablate", 6)
                ud = expand(Bx * P, segs, n, P, B=Bx)[0]
                L.llie_tune(b"pwx_ablate", 0)
                un, uo, nb, fl = expand(Bx * P, segs, n, P, B=Bx)
                tot_n += un; tot_o += uo
                print(f"B={Bx} {name:24s} pwx {un:7.1f} us {nb / un / 1e3:5.0f} GB/s {fl / un / 1e6:4.0f} TF (register-direct stores {ud:7.1f} us) | tile kernel {uo:7.1f} us {nb / uo / 1e3:5.0f} GB/s {fl / uo / 1e6:4.0f} TF", flush=True)
        print(f"sum: pwx {tot_n / 2:.1f} us, tile kernel {tot_o / 2:.1f} us")
    if "expand_nbw" in sys.argv[1:]:  # 32-channel blocks per weight buffer (LDS footprint vs barriers)
        for rep in range(2):
            for name, kind, P, segs, n in SHAPES + [("enc3.0 K1 128->512", "k1", 1024, [128], 512)]:
                if kind != "k1" or sum(segs) not in (128, 192, 256):
                    continue
                row = []
                for nbw in (1, 2, 4):
                    if nbw == 4 and sum(segs) != 128:
                        continue
                    L.llie_tune(b"pwx_nbw", nbw)
                    for abl in (0, 6):
                        L.llie_tune(b"pwx_ablate", abl)
                        row.append(f"nbw{nbw} {'tile' if abl == 0 else 'regs'} {expand(B * P, segs, n, P)[0]:6.1f}us")
                L.llie_tune(b"pwx_nbw", 0); L.llie_tune(b"pwx_ablate", 0)
                print(f"{name:22s} " + " | ".join(row), flush=True)
    if "expand_diag" in sys.argv[1:]:  # where the activation-stationary kernel spends its time (two representative shapes)
        out = (C.c_double * 4)()
        for name, P, segs, n in [("dec2.0 K1 192->768", 16384, [128, 64], 768), ("dec1.0 K1 384->1536", 4096, [256, 128], 1536)]:
            for abl in (0, 6, 3, 4, 5, 1, 2):
                L.llie_tune(b"pwx_ablate", abl)
                L.llie_tune(b"pwx_stamp", 0)
                un = min(expand(B * P, segs, n, P)[0] for _ in range(2))
                if abl > 2:  # no stamped build of these
                    print(f"{name:22s} ablate {abl}: {un:7.1f} us", flush=True)
                    continue
                L.llie_tune(b"pwx_stamp", 1)
                us = expand(B * P, segs, n, P)[0]
                N.check(L.llie_debug_pwx_stamps(out))
                print(f"{name:22s} ablate {abl}: {un:7.1f} us ({us:7.1f} stamped)  per wave: A phase {out[0]:8.0f} cyc, channel loop {out[1]:8.0f} cyc "
                      f"of which {out[2]:8.0f} in the DMA / store wait ({out[3]:.0f} waves)", flush=True)
            L.llie_tune(b"pwx_ablate", 0)
            L.llie_tune(b"pwx_stamp", 0)
    if "gemm_stamp" in sys.argv[1:]:
        out = (C.c_double * 3)()
        for name, kind, P, segs, n in SHAPES:
            L.llie_tune(b"gemm_stamp", 0)
            us0, _, _ = gemm(kind, B * P, segs, n, P)
            L.llie_tune(b"gemm_stamp", 1)
            us, gbs, tf = gemm(kind, B * P, segs, n, P)
            N.check(L.llie_debug_gemm_stamps(out))
            tot = out[0] + out[1]
            print(f"{name:28s} {us0:8.1f} us ({us:8.1f} stamped)  per wave: K loop {out[0]:9.0f} cyc ({100*out[0]/max(tot,1):4.1f}%)  epilogue {out[1]:9.0f} cyc "
                  f"({100*out[1]/max(tot,1):4.1f}%)  waves {out[2]:.0f}", flush=True)
        L.llie_tune(b"gemm_stamp", 0)
    if "bk128" in sys.argv[1:]:  # 128-wide K chunks; usage: gpu_tune.py bk128 [B]
        i = sys.argv.index("bk128")
        Bx = int(sys.argv[i + 1]) if len(sys.argv) > i + 1 else 32
        for name, kind, P, segs, n in SHAPES + [("attn qkv 256->768", "k1", 1024, [256], 768), ("attn out 256->256", "k3", 1024, [256], 256), ("enc3.0 K3 512->256", "k3", 1024, [512], 256)]:
            if n % 128 or any(c % 128 for c in segs):
                continue
            row = []
            for knob in (0, 1 << 30, 0, 1 << 30):
                L.llie_tune(b"gemm_bk128", knob)
                us, gbs, tf = gemm(kind, Bx * P, segs, n, P, act=3 if kind == "k1" else 1, B=Bx)
                row.append(f"bk={128 if knob else 64}: {us:6.1f}us")
            print(f"B={Bx} grid={(Bx * P // 128) * (n // 128):6d} {name:28s} " + " | ".join(row), flush=True)
        L.llie_tune(b"gemm_bk128", 0)
    if "xcd" in sys.argv[1:]:  # XCD-aware tile order of the production kernel (gemm_ablate bit 4)
        for name, kind, P, segs, n in SHAPES:
            row = []
            for knob in (0, 16, 0, 16):
                L.llie_tune(b"gemm_ablate", knob)
                us, gbs, tf = gemm(kind, B * P, segs, n, P, act=3 if kind == "k1" else 1)
                row.append(f"xcd={knob >> 4}: {us:7.1f}us {tf:4.0f}TF {gbs:5.0f}GB/s")
            print(f"{name:28s} " + " | ".join(row), flush=True)
        L.llie_tune(b"gemm_ablate", 0)
    if "one" in sys.argv[1:]:  # one shape, for PMC passes: gpu_tune.py one <shape index>
        i = sys.argv.index("one")
        name, kind, P, segs, n = SHAPES[int(sys.argv[i + 1])]
        us, gbs, tf = gemm(kind, B * P, segs, n, P, act=3 if kind == "k1" else 1)
        print(f"{name}: {us:.1f} us {tf:.0f} TF {gbs:.0f} GB/s")
    if "dw" in sys.argv[1:]:
        for rep in range(2):
            for H, Cc in [(256, 128), (256, 384), (128, 768), (64, 1536)]:
                row = []
                for mode in (0, 1):
                    L.llie_tune(b"dw_swap", mode)
                    us, gbs = dw(B, H, Cc)
                    row.append(f"swap{mode}: {us:7.1f}us {gbs:5.0f}GB/s")
                print(f"dw {H}x{H} C={Cc}: " + " | ".join(row), flush=True)
        L.llie_tune(b"dw_swap", 0)


def chain(B, Bc, P, H, cin, hid, cout, iters=5):
    """K1 -> depthwise -> K3 of one inverted-residual block over B images in chunks of Bc (fp16):
    does keeping a chunk's hidden tensors in the Infinity Cache beat full-batch launches?"""
    tdt = torch.float16
    x = torch.randn(B * P, cin, device=dev, dtype=tdt)
    w1 = torch.randn(hid, cin, device=dev, dtype=tdt) * 0.05
    w3 = torch.randn(cout, hid, device=dev, dtype=tdt) * 0.05
    h1 = torch.empty(Bc * P, hid, device=dev, dtype=tdt)
    h2 = torch.empty(Bc * P, hid, device=dev, dtype=tdt)
    y = torch.empty(B * P, cout, device=dev, dtype=tdt)
    sc = torch.rand(B, max(cin, hid), device=dev) + 0.5
    bi = torch.randn(B, max(cin, hid), device=dev) * 0.1
    wd = torch.randn(9, hid, device=dev) * 0.3
    tiles = P // L.llie_pw_gemm_tile_rows(P)
    st1 = torch.empty(B * tiles * 2 * hid, device=dev)
    st3 = torch.empty(B * tiles * 2 * cout, device=dev)
    pool = torch.empty(B * L.llie_dwconv3x3_tiles(H, H) * hid, device=dev)
    es = 2

    def run():
        for b0 in range(0, B, Bc):
            xo = x.data_ptr() + b0 * P * cin * es
            s1 = (N.GemmSeg * 1)(N.GemmSeg(xo, cin, sc.data_ptr() + b0 * sc.shape[1] * 4, bi.data_ptr() + b0 * sc.shape[1] * 4, sc.shape[1], 1))
            N.check(L.llie_pw_gemm(1, s1, 1, w1.data_ptr(), None, None, h1.data_ptr(), st1.data_ptr(), Bc * P, hid, P, st()))
            N.check(L.llie_dwconv3x3(1, h1.data_ptr(), h2.data_ptr(), sc.data_ptr() + b0 * sc.shape[1] * 4, bi.data_ptr() + b0 * sc.shape[1] * 4,
                                     wd.data_ptr(), pool.data_ptr(), Bc, H, H, hid, st()))
            s3 = (N.GemmSeg * 1)(N.GemmSeg(h2.data_ptr(), hid, sc.data_ptr() + b0 * sc.shape[1] * 4, None, sc.shape[1], 0))
            N.check(L.llie_pw_gemm(1, s3, 1, w3.data_ptr(), None, xo if cin == cout else None,
                                   y.data_ptr() + b0 * P * cout * es, st3.data_ptr(), Bc * P, cout, P, st()))
    return time_it(run, iters)


if __name__ == "__main__" and "chain" in sys.argv[1:]:
    for name, P, H, cin, hid, cout in [("enc0 32-128-32 @256", 65536, 256, 32, 128, 32), ("dec3.0 96-384-32 @256", 65536, 256, 96, 384, 32),
                                        ("enc1.1 64-256-64 @128", 16384, 128, 64, 256, 64), ("dec2.0 192-768-64 @128", 16384, 128, 192, 768, 64),
                                        ("mid 256-1024-256 @32", 1024, 32, 256, 1024, 256)]:
        row = []
        for rep in range(2):
            for Bc in (32, 16, 8, 4, 2):
                us = chain(32, Bc, P, H, cin, hid, cout)
                row.append(f"Bc={Bc}: {us:7.0f}us")
        print(f"{name:26s} " + " | ".join(row), flush=True)
