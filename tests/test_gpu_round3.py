"""GPU parity tests added in round 3 (through the C ABI).

  * the activation-stationary expand GEMM (pwx.hip, llie_pw_expand) against a plain PyTorch fp32 restatement of the same
    operator (efficient_unet.py:207-208 norm1 + ReLU6 prologue, :174 expand, the statistics norm2 :212 needs), against the
    tile kernel it replaces (llie_pw_gemm), bitwise batch invariance, and through the whole network (knob "pwx")
  * a batch equals its halves bit for bit whatever the launch-size heuristics choose (fp32 / fp16 / bf16, both expand paths)
  * GroupNorm-2 statistics of the recompute form from the Gram matrix of the block input (gram.hip, llie_gram_stats) against a
    float64 restatement, and through the whole network with the knob on and off
  * the remaining per-kernel entry points of SURVEY.md 8b (llie_groupnorm_finalize, llie_conv3x3, llie_linattn, llie_se_mlp,
    llie_film) against plain PyTorch
"""
import importlib
import math

import numpy as np
import pytest
import torch

import oracle
from conftest import synth_input

pytestmark = pytest.mark.gpu
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _expand_case(dev, tdt, segs, nout, P, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    K = sum(segs)
    xs = [(torch.randn(B * P, c, generator=g) * 1.5).to(tdt) for c in segs]
    w32 = torch.randn(nout, K, generator=g) / math.sqrt(K)
    # tables as gn_finalize emits them for ACT_RELU6_S6: scale / 6 and shift / 6 per (image, channel)
    sc = (torch.rand(B, K, generator=g) + 0.5) / 6
    bi = (torch.randn(B, K, generator=g) * 0.5 + 0.4) / 6
    return xs, w32, sc, bi


def _run_expand(dev, dtype, tdt, xs, w32, sc, bi, nout, P, B):
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    K = w32.shape[1]
    xd = [x.to(dev) for x in xs]
    scd, bid, wd = sc.to(dev), bi.to(dev), w32.to(dev)
    out = torch.full((B * P, nout), float("nan"), dtype=tdt, device=dev)
    wpack = torch.empty(nout * K, dtype=tdt, device=dev)
    stats = torch.full((B, P // 128, 2, nout), float("nan"), device=dev)
    arr = (N.GemmSeg * len(xs))()
    off = 0
    for i, x in enumerate(xd):
        arr[i] = N.GemmSeg(x.data_ptr(), x.shape[1], scd.data_ptr() + off * 4, bid.data_ptr() + off * 4, K, 3)
        off += x.shape[1]
    N.check(L.llie_pw_expand(dtype, arr, len(xs), wd.data_ptr(), wpack.data_ptr(), out.data_ptr(), stats.data_ptr(),
                             B * P, nout, P, st), "pw_expand")
    # the tile kernel on the same operands (weights rounded to T, x 6 in its epilogue)
    out2 = torch.empty_like(out)
    stats2 = torch.empty_like(stats)
    wt = wd.to(tdt)
    N.check(L.llie_pw_gemm(dtype, arr, len(xs), wt.data_ptr(), None, None, out2.data_ptr(), stats2.data_ptr(), B * P, nout, P, st), "pw_gemm")
    torch.cuda.synchronize()
    return out.cpu(), stats.cpu(), out2.cpu(), stats2.cpu()


@pytest.mark.parametrize("dtype,tdt,ulp", [(1, torch.float16, 2.0 ** -10), (2, torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("segs,nout,P,B", [([128], 512, 256, 3), ([128, 64], 768, 128, 2), ([256], 1024, 1024, 2),
                                           ([256, 128], 1536, 256, 1), ([256, 256], 2048, 128, 2), ([128], 512, 4096, 2)])
def test_pw_expand_vs_torch_and_tile_kernel(dev, dtype, tdt, ulp, segs, nout, P, B):
    xs, w32, sc, bi = _expand_case(dev, tdt, segs, nout, P, B, seed=len(segs) * 1000 + nout)
    out, stats, out2, stats2 = _run_expand(dev, dtype, tdt, xs, w32, sc, bi, nout, P, B)
    assert torch.isfinite(out.float()).all() and torch.isfinite(stats).all()
    # plain PyTorch restatement (fp32 on the operands as the kernel sees them: activation rounded once to T, weights = T(6 W))
    x = torch.cat([v.float() for v in xs], 1).view(B, P, -1)
    a = (x.double() * sc[:, None, :].double() + bi[:, None, :].double()).clamp(0, 1).to(tdt).double()
    w6 = (w32 * 6).to(tdt).double()
    ref = (a @ w6.t()).view(B * P, nout)
    err = (out.double() - ref).abs()
    # one rounding to T of an fp32-accumulated sum, plus the rare activation whose own rounding to T falls the other way
    # (fp32 FMA on the device, float64 here: one unit in the last place of a' <= 1 times one weight, |6 W| < 2.5)
    tol = ulp * ref.abs() + ulp * 2.5
    assert (err <= tol).all(), (err / tol).max().item()
    assert (err.norm() / ref.norm()).item() < ulp / 2
    # statistics: exactly the sums of the values that were stored (fp32 summation order aside)
    o = out.double().view(B, P // 128, 128, nout)
    s_ref = torch.stack([o.sum(2), (o * o).sum(2)], 2)
    assert torch.allclose(stats.double(), s_ref, rtol=2e-6, atol=1e-3), (stats.double() - s_ref).abs().max().item()
    # the kernel it replaces computes T(6 * (a . T(W))): same values up to the weights' rounding
    rel = (out.double() - out2.double()).norm() / out2.double().norm()
    assert rel < 4 * ulp, rel.item()
    assert torch.allclose(stats[:, :, 0].double(), stats2[:, :, 0].double(), rtol=0.02, atol=0.3 * math.sqrt(128) * max(1.0, float(out2.float().abs().max())) * ulp * 8)


def test_pw_expand_is_bitwise_batch_invariant(dev):
    tdt, P, nout, segs = torch.float16, 1024, 1024, [256]
    xs, w32, sc, bi = _expand_case(dev, tdt, segs, nout, P, 4, seed=5)
    out4, st4, _, _ = _run_expand(dev, 1, tdt, xs, w32, sc, bi, nout, P, 4)
    for b in (0, 3):  # a single image alone (different grid, different nsplit) gives the same bits
        out1, st1, _, _ = _run_expand(dev, 1, tdt, [x[b * P:(b + 1) * P] for x in xs], w32, sc[b:b + 1], bi[b:b + 1], nout, P, 1)
        assert torch.equal(out1, out4[b * P:(b + 1) * P]) and torch.equal(st1[0], st4[b])
    again, st_again, _, _ = _run_expand(dev, 1, tdt, xs, w32, sc, bi, nout, P, 4)
    assert torch.equal(again, out4) and torch.equal(st_again, st4)


def test_pw_expand_refuses_shapes_outside_its_contract(dev):
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(256, 96, dtype=torch.float16, device=dev)
    t = torch.zeros(1, 96, device=dev)
    w = torch.zeros(384, 96, device=dev)
    buf = torch.zeros(384 * 256, dtype=torch.float16, device=dev)
    stats = torch.zeros(2 * 2 * 384, device=dev)
    arr = (N.GemmSeg * 1)(N.GemmSeg(x.data_ptr(), 96, t.data_ptr(), t.data_ptr(), 96, 3))
    assert L.llie_pw_expand(1, arr, 1, w.data_ptr(), buf.data_ptr(), buf.data_ptr(), stats.data_ptr(), 256, 384, 128, st) == -2  # K = 96
    x2 = torch.zeros(192, 128, dtype=torch.float16, device=dev)
    arr = (N.GemmSeg * 1)(N.GemmSeg(x2.data_ptr(), 128, t.data_ptr(), t.data_ptr(), 128, 3))
    assert L.llie_pw_expand(1, arr, 1, w.data_ptr(), buf.data_ptr(), buf.data_ptr(), stats.data_ptr(), 192, 512, 96, st) == -2  # P % 128
    arr = (N.GemmSeg * 1)(N.GemmSeg(x2.data_ptr(), 128, t.data_ptr(), t.data_ptr(), 128, 1))
    assert L.llie_pw_expand(1, arr, 1, w.data_ptr(), buf.data_ptr(), buf.data_ptr(), stats.data_ptr(), 128, 512, 128, st) == -2  # act != 3
    assert L.llie_pw_expand(0, arr, 1, w.data_ptr(), buf.data_ptr(), buf.data_ptr(), stats.data_ptr(), 128, 512, 128, st) == -1   # fp32


def _psnr01(a, b):
    a = (a.double().clamp(-1, 1) + 1) / 2
    b = (b.double().clamp(-1, 1) + 1) / 2
    mse = ((a - b) ** 2).mean().item()
    return 99.0 if mse == 0 else 10 * math.log10(1.0 / mse)


@pytest.mark.parametrize("cd,min_psnr", [("fp16", 50.0), ("bf16", 32.0)])
def test_whole_network_with_and_without_activation_stationary_expand(dev, cd, min_psnr):
    """small@128 (wide blocks at 32 x 32 and 16 x 16 pixels take the new kernel; the 16 x 16 level has P = 256): against
    the fp32 CPU oracle by PSNR with the knob on and off, and the two engines against each other."""
    L = N.lib()
    spec = oracle.make_spec("small", 128)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    m = M.LowLightDiffusion(unet_variant="small", image_size=128, num_inference_steps=4, compute_dtype=cd)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    low = synth_input("r3:low128", (3, 3, 128, 128), -1.0, -0.4)
    noise = oracle.draw_noise(3, 128, 4, seed=21)
    ref = oracle.enhance_ref(sd, spec, low, 4, noise)["enhanced"]
    outs = {}
    try:
        for knob in (1, 0):
            N.check(L.llie_tune(b"pwx", knob))
            outs[knob] = m.enhance(low.to(dev), 4, noise=torch.stack(noise)).cpu()
            assert _psnr01(outs[knob], ref) >= min_psnr, (knob, _psnr01(outs[knob], ref))
    finally:
        L.llie_tune(b"pwx", 1)
    assert _psnr01(outs[1], outs[0]) >= min_psnr
    # sub-batches stay bitwise identical with the new kernel in the path
    one = m.enhance(low[1:2].to(dev), 4, noise=torch.stack(noise)[:, 1:2]).cpu()
    assert torch.equal(one[0], outs[1][1])


@pytest.mark.parametrize("cd,pwx", [("fp16", 1), ("fp16", 0), ("bf16", 0), (None, 1)])
def test_batch_equals_its_halves_whatever_the_grid_heuristics_choose(dev, cd, pwx):
    """small@64 with B = 6 against B = 3 + 3: launch-size heuristics (K chunk width of pw_gemm, channel split of pw_expand)
    fall on different sides of their thresholds for the two batch sizes on several layers; outputs must not differ in any bit
    (a round-3 heuristic that narrowed pw_gemm's N tiles on small grids changed the statistics' summation order and was
    removed for failing exactly this).  pwx = 0 sends the wide expands through pw_gemm too; the fp32 engine always does."""
    L = N.lib()
    spec = oracle.make_spec("small", 64)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    m = M.LowLightDiffusion(unet_variant="small", image_size=64, num_inference_steps=4, compute_dtype=cd)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    low = synth_input("r3:low64b", (6, 3, 64, 64), -1.0, -0.4).to(dev)
    noise = torch.stack(oracle.draw_noise(6, 64, 4, seed=5)).to(dev)
    try:
        N.check(L.llie_tune(b"pwx", pwx))
        full = m.enhance(low, 4, noise=noise, return_noise_pred=True)
        parts = [m.enhance(low[a:b], 4, noise=noise[:, a:b], return_noise_pred=True) for a, b in ((0, 3), (3, 6))]
        one = m.enhance(low[4:5], 4, noise=noise[:, 4:5])
    finally:
        L.llie_tune(b"pwx", 1)
    for i in range(4):
        assert torch.equal(torch.cat([q.noise_pred[i] for q in parts]), full.noise_pred[i]), i
    assert torch.equal(torch.cat([q.enhanced for q in parts]), full.enhanced)
    assert torch.equal(one, full.enhanced[4:5])


# ------------------------------------------------------------------ GroupNorm-2 statistics from the Gram matrix (gram.hip)
@pytest.mark.parametrize("dtype,tdt", [(1, torch.float16), (2, torch.bfloat16)])
@pytest.mark.parametrize("c0,c1,P,B", [(32, 0, 4096, 3), (32, 32, 16384, 2), (64, 32, 8192, 2), (64, 0, 512, 3), (16, 16, 1024, 2), (96, 0, 65536, 1)])
def test_gram_stats_vs_torch(dev, dtype, tdt, c0, c1, P, B):
    """llie_gram_stats against a float64 restatement: G = sum_px a a^T, m = sum_px a of a = clamp01(x * s + b) rounded to the
    compute type (the operand relu6(norm1(x)) / 6 of the expand GEMM, efficient_unet.py:207-208); concat inputs, one to 512
    workgroup partials per image; a single image alone gives the same bits; the tickets are left zero."""
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    K = c0 + c1
    g = torch.Generator().manual_seed(K * 1000 + P)
    x0 = (torch.randn(B, P, c0, generator=g) * 1.5).to(tdt)
    x1 = (torch.randn(B, P, c1, generator=g) * 1.5).to(tdt) if c1 else None
    sc = (torch.rand(B, K, generator=g) + 0.5) / 6
    bi = (torch.randn(B, K, generator=g) * 0.5 + 0.4) / 6

    def run(sl):
        nb = sl.stop - sl.start
        d0 = x0[sl].contiguous().to(dev)
        d1 = x1[sl].contiguous().to(dev) if c1 else None
        s_d, b_d = sc[sl].contiguous().to(dev), bi[sl].contiguous().to(dev)
        part = torch.full((nb * int(L.llie_gram_part_floats(K, P)),), float("nan"), device=dev)
        gtot = torch.full((nb, K * K + K), float("nan"), device=dev)
        tick = torch.zeros(nb, dtype=torch.int32, device=dev)
        for _ in range(2):  # the second launch runs on the tickets the first one left
            N.check(L.llie_gram_stats(dtype, d0.data_ptr(), c0, d1.data_ptr() if c1 else None, c1, s_d.data_ptr(), b_d.data_ptr(), nb, P,
                                      part.data_ptr(), gtot.data_ptr(), tick.data_ptr(), st), "gram_stats")
        torch.cuda.synchronize()
        assert int(tick.abs().sum()) == 0
        return gtot.cpu()

    got = run(slice(0, B))
    assert torch.isfinite(got).all()
    x = torch.cat([x0, x1], 2) if c1 else x0
    a = (x.double() * sc[:, None, :].double() + bi[:, None, :].double()).clamp(0, 1).to(tdt).double()
    G = torch.einsum("bpi,bpj->bij", a, a)
    mref = a.sum(1)
    ulp = 2.0 ** -10 if tdt == torch.float16 else 2.0 ** -7
    # fp32 accumulation of exact products, plus the rare activation that rounds to T the other way on the device (fp32 FMA
    # there, float64 here): one unit in the last place of one factor of one of P terms
    gg = got[:, :K * K].view(B, K, K).double()
    assert (gg - G).abs().max() <= 2e-5 * G.abs().max() + 4 * ulp, ((gg - G).abs().max().item(), G.abs().max().item())
    assert torch.equal(gg, gg.transpose(1, 2))
    mm = got[:, K * K:].double()
    assert (mm - mref).abs().max() <= 2e-5 * mref.abs().max() + 4 * ulp
    one = run(slice(B - 1, B))
    assert torch.equal(one[0], got[B - 1])


@pytest.mark.parametrize("cd,min_psnr", [("fp16", 50.0), ("bf16", 32.0)])
def test_whole_network_with_gram_statistics_and_with_the_second_expand_pass(dev, cd, min_psnr):
    """small@128: every recompute block (Cin 32 / 64 / 96) takes its norm2 statistics from the Gram matrix (knob "gram" = 1)
    or from expand_stats (0): both against the fp32 CPU oracle by PSNR and against each other; sub-batches bit-identical."""
    L = N.lib()
    spec = oracle.make_spec("small", 128)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    m = M.LowLightDiffusion(unet_variant="small", image_size=128, num_inference_steps=4, compute_dtype=cd)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    low = synth_input("r3:low128g", (3, 3, 128, 128), -1.0, -0.4)
    noise = oracle.draw_noise(3, 128, 4, seed=22)
    ref = oracle.enhance_ref(sd, spec, low, 4, noise)["enhanced"]
    outs = {}
    try:
        for knob in (2, 0):  # 2: also below 32 768 pixels per image (every level of this 128 x 128 network)
            N.check(L.llie_tune(b"gram", knob))
            outs[knob] = m.enhance(low.to(dev), 4, noise=torch.stack(noise)).cpu()
            assert _psnr01(outs[knob], ref) >= min_psnr, (knob, _psnr01(outs[knob], ref))
        assert _psnr01(outs[2], outs[0]) >= min_psnr
        N.check(L.llie_tune(b"gram", 2))
        one = m.enhance(low[1:2].to(dev), 4, noise=torch.stack(noise)[:, 1:2]).cpu()
        assert torch.equal(one[0], outs[2][1])
        again = m.enhance(low.to(dev), 4, noise=torch.stack(noise)).cpu()
        assert torch.equal(again, outs[2])
    finally:
        L.llie_tune(b"gram", 1)


# ------------------------------------------------------------------ the remaining per-kernel entry points
def _nhwc(x, tdt):  # [B][C][H][W] fp32 -> [B][H*W][C] T
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(b, h * w, c).contiguous().to(tdt)


@pytest.mark.parametrize("c0,c1,P,film", [(32, 0, 4096, False), (64, 32, 1024, True), (256, 0, 256, True)])
def test_groupnorm_finalize_entry_point_vs_torch(dev, c0, c1, P, film):
    """llie_groupnorm_finalize: slabs of per-tile (sum, sum of squares) -> the affine a consumer applies on load; against
    F.group_norm (efficient_unet.py:170-171) with FiLM folded in (:215-217), over a virtual concat of two tensors."""
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, C, groups = 3, c0 + c1, 32
    g = torch.Generator().manual_seed(C + P)
    y = torch.randn(B, P, C, generator=g) * 1.7 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    fl = torch.randn(B, 2 * C, generator=g) * 0.3
    nt = P // 128
    def slab(t):
        t = t.view(B, nt, 128, t.shape[-1]).double()
        return torch.stack([t.sum(2), (t * t).sum(2)], 2).float().contiguous().to(dev)
    s0 = slab(y[..., :c0].contiguous())
    s1 = slab(y[..., c0:].contiguous()) if c1 else None
    sc, sh = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
    gd, bd, fd = gamma.to(dev), beta.to(dev), fl.to(dev)
    N.check(L.llie_groupnorm_finalize(s0.data_ptr(), nt, c0, s1.data_ptr() if c1 else None, nt if c1 else 0, c1, groups, P, gd.data_ptr(), bd.data_ptr(),
                                      fd.data_ptr() if film else None, 2 * C if film else 0, 1e-5, 0.0, B, sc.data_ptr(), sh.data_ptr(), st), "gn_finalize")
    torch.cuda.synchronize()
    got = y * sc.cpu()[:, None, :] + sh.cpu()[:, None, :]
    ref = torch.nn.functional.group_norm(y.permute(0, 2, 1).double(), groups, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1)
    if film:
        ref = ref * (1 + fl[:, None, :C].double()) + fl[:, None, C:].double()
    assert (got.double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("dtype,tdt,tol", [(0, torch.float32, 2e-5), (1, torch.float16, 4e-3), (2, torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("mode,cin,cout,hw", [(0, 32, 64, 32), (1, 64, 64, 16), (1, 128, 128, 8)])
def test_conv3x3_entry_point_vs_torch(dev, dtype, tdt, tol, mode, cin, cout, hw):
    """llie_conv3x3 against F.conv2d: Downsample (efficient_unet.py:367) and Upsample (:383-384, bilinear x2 first)."""
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    B = 2
    g = torch.Generator().manual_seed(cin + hw + mode)
    x = torch.randn(B, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g) * 0.1
    xd = _nhwc(x, tdt).to(dev)
    wd = w.permute(2, 3, 0, 1).reshape(9, cout, cin).contiguous().to(tdt).to(dev)
    bd = b.to(dev)
    ho = hw // 2 if mode == 0 else hw * 2
    out = torch.full((B, ho * ho, cout), float("nan"), dtype=tdt, device=dev)
    nt = int(L.llie_conv3x3_tiles(ho, ho))
    stats = torch.full((B, nt, 2, cout), float("nan"), device=dev)
    N.check(L.llie_conv3x3(dtype, mode, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), stats.data_ptr(), B, hw, hw, cin, cout, st), "conv3x3")
    torch.cuda.synchronize()
    xr, wr = xd.cpu().float().view(B, hw, hw, cin).permute(0, 3, 1, 2), wd.cpu().float().view(3, 3, cout, cin).permute(2, 3, 0, 1)
    if mode == 0:
        ref = torch.nn.functional.conv2d(xr.double(), wr.double(), b.double(), stride=2, padding=1)
    else:
        up = torch.nn.functional.interpolate(xr.double(), scale_factor=2, mode="bilinear", align_corners=False)
        ref = torch.nn.functional.conv2d(up, wr.double(), b.double(), padding=1)
    got = out.cpu().double().view(B, ho, ho, cout).permute(0, 3, 1, 2)
    assert (got - ref).abs().max() < tol * max(1.0, ref.abs().max().item()), (got - ref).abs().max().item()
    o = out.cpu().double()
    assert torch.allclose(stats.cpu().double().sum(1)[:, 0], o.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(stats.cpu().double().sum(1)[:, 1], (o * o).sum(1), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("dtype,tdt,tol", [(0, torch.float32, 2e-5), (1, torch.float16, 4e-3), (2, torch.bfloat16, 3e-2)])
def test_linattn_entry_point_vs_torch(dev, dtype, tdt, tol):
    """llie_linattn against the restatement of LinearAttention's core (efficient_unet.py:288-302; phi = elu + 1, no scale)."""
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, n, heads = 2, 1024, 4
    inner = heads * 32
    g = torch.Generator().manual_seed(7)
    qkv = (torch.randn(B, n, 3 * inner, generator=g) * 0.8).to(tdt)
    qd = qkv.to(dev)
    kv = torch.empty(int(L.llie_linattn_splits(n)) * B * heads * 32 * 33, device=dev)
    out = torch.full((B, n, inner), float("nan"), dtype=tdt, device=dev)
    N.check(L.llie_linattn(dtype, qd.data_ptr(), kv.data_ptr(), out.data_ptr(), B, n, heads, st), "linattn")
    torch.cuda.synchronize()
    q, k, v = (z.double().view(B, n, heads, 32).permute(0, 2, 3, 1) for z in qkv.split(inner, dim=2))  # [b][h][d][n]
    q, k = torch.nn.functional.elu(q) + 1, torch.nn.functional.elu(k) + 1
    kvr = torch.einsum("bhdn,bhen->bhde", k, v)
    num = torch.einsum("bhdn,bhde->bhen", q, kvr)
    den = torch.einsum("bhdn,bhd->bhn", q, k.sum(-1))[:, :, None, :] + 1e-6
    ref = (num / den).permute(0, 3, 1, 2).reshape(B, n, inner)
    assert (out.cpu().double() - ref).abs().max() < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,tdt,tol", [(0, torch.float32, 1e-5), (1, torch.float16, 2e-3), (2, torch.bfloat16, 1.5e-2)])
def test_se_mlp_and_film_entry_points_vs_torch(dev, dtype, tdt, tol):
    """llie_se_mlp (SqueezeExcitation, efficient_unet.py:96-100) and llie_film (:189-192) against plain PyTorch."""
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, C, Cs, P = 5, 512, 128, 4096
    g = torch.Generator().manual_seed(11)
    sums = torch.randn(B, C, generator=g) * P * 0.3
    w1 = (torch.randn(Cs, C, generator=g) / math.sqrt(C)).to(tdt)
    w2 = (torch.randn(C, Cs, generator=g) / math.sqrt(Cs)).to(tdt)
    b1, b2 = torch.randn(Cs, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    d = [t.to(dev) for t in (sums, w1, b1, w2, b2)]
    mean, hid, gate = torch.empty(B, C, device=dev), torch.empty(B, Cs, device=dev), torch.full((B, C), float("nan"), device=dev)
    N.check(L.llie_se_mlp(dtype, d[0].data_ptr(), P, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), mean.data_ptr(), hid.data_ptr(),
                          gate.data_ptr(), B, C, Cs, st), "se_mlp")
    m = sums.double() / P
    h = torch.nn.functional.relu6(m @ w1.double().t() + b1.double())
    ref = torch.sigmoid(h @ w2.double().t() + b2.double())
    torch.cuda.synchronize()
    assert (gate.cpu().double() - ref).abs().max() < tol
    rows, T, F_ = 3, 128, 1000
    stemb = torch.randn(rows, T, generator=g)
    wf, bf = torch.randn(F_, T, generator=g) / math.sqrt(T), torch.randn(F_, generator=g) * 0.1
    film = torch.full((rows, F_), float("nan"), device=dev)
    dd = [t.to(dev) for t in (stemb, wf, bf)]
    N.check(L.llie_film(dd[0].data_ptr(), dd[1].data_ptr(), dd[2].data_ptr(), film.data_ptr(), rows, T, F_, st), "film")
    torch.cuda.synchronize()
    assert (film.cpu().double() - (stemb.double() @ wf.double().t() + bf.double())).abs().max() < 2e-5
