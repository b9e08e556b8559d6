"""GPU tests added in round 4 (through the C ABI).

  * llie_gram_finalize (gram.hip: gram_finalize_kernel) against a float64 restatement of GroupNorm-2 + FiLM over h1 = 6 W1 a',
    including an ill-conditioned case (a' with a large mean and a small variance, weight rows that sum to ~0)
  * launch-policy knobs that must never change a bit: the cache policy of the big tensors' stores ("nt_mask" / "nt_min_mb"),
    expand_dw's grid ("irbx_grid*"): whole network, every compute dtype of the 2-byte engines
  * the hipGraph cache of llie_enhance is bounded (least recently used entry evicted) and an evicted key still gives its bits
  * SqueezeExcitation MLP of the wide blocks on the matrix pipe (se_fc1_mfma / se_fc2_mfma) vs the oracle and the row-parallel pair
  * llie_pw_gemm with K segments of 64 n + 32 channels: 64-wide chunks with half-empty segment tails, bit for bit the 32-wide chunks
  * llie_optimizer_step (FusedAdamW: clip_grad_norm_ + AdamW + EMA in three launches) vs torch.optim.AdamW, skip of a non-finite
    step, checkpoint layout in both directions, LR scheduler; TrainStep vs the autograd path
"""
import importlib
import math

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


# ------------------------------------------------------------------ gram_finalize
def _gram_case(K, P, B, tdt, hard, seed):
    g = torch.Generator().manual_seed(seed)
    if hard:  # nearly constant activations near the clamp's upper end, expand rows with (almost) zero sum
        a = (0.9 + 0.02 * torch.randn(B, P, K, generator=g)).clamp(0, 1)
        w = torch.randn(4 * K, K, generator=g) / math.sqrt(K)
        w = w - w.mean(1, keepdim=True) * 0.98
    else:
        a = torch.rand(B, P, K, generator=g).clamp(0, 1) * (torch.rand(B, 1, K, generator=g) + 0.2)
        w = torch.randn(4 * K, K, generator=g) / math.sqrt(K)
    return a.to(tdt), w.to(tdt)


@pytest.mark.parametrize("dtype,tdt", [(1, torch.float16), (2, torch.bfloat16)])
@pytest.mark.parametrize("K,P,hard", [(32, 4096, False), (64, 2048, False), (96, 8192, False), (32, 65536, True), (96, 16384, True)])
def test_gram_finalize_entry_point_vs_float64(dev, dtype, tdt, K, P, hard):
    """scale / shift of norm2 + FiLM (efficient_unet.py:212-217) from (G, m): against float64 statistics of h1 = 6 W a' taken
    over the pixels themselves.  G and m are handed over in fp32 as llie_gram_stats leaves them (rounded from float64 here), so
    the bound covers the finalize and the fp32 storage of G -- in the ill-conditioned case that storage dominates: the
    variance is recovered from a quadratic form whose terms are ~1e4 times larger than the result."""
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, C = 2, 4 * K
    a, w = _gram_case(K, P, B, tdt, hard, seed=K + P)
    ad, wd = a.double(), w.double()
    G = torch.einsum("bpi,bpj->bij", ad, ad)
    m = ad.sum(1)
    gtot = torch.cat([G.reshape(B, K * K), m], 1).float().contiguous().to(dev)
    g = torch.Generator().manual_seed(7)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    fl = torch.randn(B, 2 * C, generator=g) * 0.3
    sc, sh = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
    wdev, gd, bd, fd = w.to(dev), gamma.to(dev), beta.to(dev), fl.to(dev)
    N.check(L.llie_gram_finalize(dtype, gtot.data_ptr(), wdev.data_ptr(), K, P, gd.data_ptr(), bd.data_ptr(), fd.data_ptr(), 2 * C, 1e-5, 0.0,
                                 B, sc.data_ptr(), sh.data_ptr(), st), "gram_finalize")
    torch.cuda.synchronize()
    h = 6.0 * torch.einsum("bpk,ck->bpc", ad, wd)                      # [B][P][C]
    hg = h.view(B, P, 32, C // 32)
    mean = hg.mean((1, 3))
    var = (hg * hg).mean((1, 3)) - mean * mean
    rstd = 1.0 / torch.sqrt(var.clamp_min(0) + 1e-5)
    mean_c, rstd_c = mean.repeat_interleave(C // 32, 1), rstd.repeat_interleave(C // 32, 1)
    fs, fh = 1.0 + fl[:, :C].double(), fl[:, C:].double()
    sc_ref = rstd_c * gamma.double() * fs
    sh_ref = (beta.double() - mean_c * rstd_c * gamma.double()) * fs + fh
    # judged on what the consumer computes: the normalised tensor h * scale + shift
    got = h * sc.cpu().double()[:, None, :] + sh.cpu().double()[:, None, :]
    ref = h * sc_ref[:, None, :] + sh_ref[:, None, :]
    err = (got - ref).abs().max().item()
    assert err < (2e-3 if hard else 2e-5), err


# ------------------------------------------------------------------ launch-policy knobs never change a bit
@pytest.mark.parametrize("cd", ["fp16", "bf16"])
def test_store_policy_and_grid_knobs_do_not_change_a_bit(dev, cd):
    """small@128, B=3: non-temporal stores for every producer and every tensor size (nt_mask = 31, nt_min_mb = 1), none at all,
    and three expand_dw grid choices all give the default build's bits."""
    L = N.lib()
    spec = oracle.make_spec("small", 128)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    m = M.LowLightDiffusion(unet_variant="small", image_size=128, num_inference_steps=4, compute_dtype=cd)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    low = synth_input("r4:low128", (3, 3, 128, 128), -1.0, -0.4).to(dev)
    noise = torch.stack(oracle.draw_noise(3, 128, 4, seed=31)).to(dev)
    base = m.enhance(low, 4, noise=noise).cpu()
    settings = [{"nt_mask": 31, "nt_min_mb": 1}, {"nt_mask": 0}, {"irbx_grid": 96}, {"irbx_grid2": 7, "irbx_grid4": 1000, "irbx_grid6": 48}]
    defaults = {"nt_mask": 1, "nt_min_mb": 100, "irbx_grid": 0, "irbx_grid2": 0, "irbx_grid4": 0, "irbx_grid6": 0}
    try:
        for s in settings:
            for k, v in s.items():
                N.check(L.llie_tune(k.encode(), v))
            for _ in range(2):  # eager, then the captured graph
                out = m.enhance(low, 4, noise=noise).cpu()
                assert torch.equal(out, base), s
            for k in s:
                N.check(L.llie_tune(k.encode(), defaults[k]))
    finally:
        for k, v in defaults.items():
            L.llie_tune(k.encode(), v)


# ------------------------------------------------------------------ bounded hipGraph cache
def test_graph_cache_is_bounded_and_evicted_keys_still_work(dev):
    """20 different (batch, steps) keys on one model: at most 16 captured graphs stay alive, and the first key -- evicted by
    then -- reproduces its first result bit for bit when it comes back (eager, captured again, replayed)."""
    L = N.lib()
    m = M.LowLightDiffusion(unet_variant="small", image_size=64, num_inference_steps=4, compute_dtype="fp16").to(dev).eval()
    low = synth_input("r4:low64", (5, 3, 64, 64), -1.0, -0.4).to(dev)
    noise = torch.stack(oracle.draw_noise(5, 64, 4, seed=3)).to(dev)
    keys = [(b, n) for n in (4, 3, 2, 1) for b in (1, 2, 3, 4, 5)]
    first = {}
    h = None
    for b, n in keys:
        for rep in range(3):  # eager, capture, replay
            out = m.enhance(low[:b], n, noise=noise[:n, :b]).cpu()
            if rep == 0:
                first[(b, n)] = out
            else:
                assert torch.equal(out, first[(b, n)]), (b, n, rep)
        h = m.unet._prepare(b, dev)[0]
        assert 0 < L.llie_graph_cache_entries(h.h) <= 16
    assert L.llie_graph_cache_entries(h.h) == 16
    b, n = keys[0]
    for rep in range(3):
        assert torch.equal(m.enhance(low[:b], n, noise=noise[:n, :b]).cpu(), first[(b, n)]), rep
    assert L.llie_graph_cache_entries(h.h) == 16


# ------------------------------------------------------------------ SE MLP of the wide blocks on the matrix pipe
def _irb_case(cin, cout, hw, b, name):
    from oracle import unet_ref
    from oracle.weightgen import synth_tensor
    blk = M.InvertedResidualBlock(cin, cout, 128)
    blk.load_state_dict({k: synth_tensor(name + "." + k, tuple(v.shape)) for k, v in blk.state_dict().items()})
    sd = {name + "." + k: v.detach().clone() for k, v in blk.state_dict().items()}
    x = synth_input(name + ".x", (b, cin, hw, hw), -2, 2)
    te = synth_input(name + ".temb", (b, 128), -1, 1)
    return blk, x, te, unet_ref.irb_forward(sd, name, x, te)


@pytest.mark.parametrize("cd,tol", [("fp16", 6e-3), ("bf16", 5e-2)])
@pytest.mark.parametrize("cin,cout,hw,b", [(256, 256, 16, 3), (512, 256, 8, 35), (192, 64, 16, 2), (384, 128, 16, 33)])
def test_se_mlp_on_the_matrix_pipe(dev, cd, tol, cin, cout, hw, b):
    """SqueezeExcitation (efficient_unet.py:96-100) of blocks with 768 ... 2048 hidden channels as two MFMA launches
    (se_fc1_mfma / se_fc2_mfma: the batch is the rows of a 32 x 32 tile; batches of 33 / 35 need a second row block) against the
    fp32 CPU oracle through the whole block, against the row-parallel pair it replaces (knob "se_mfma" = 0), and bitwise
    independent of the batch: the first image alone gives the bits it has inside the batch."""
    L = N.lib()
    blk, x, te, ref = _irb_case(cin, cout, hw, b, f"r4se_{cin}_{cout}")
    blk = blk.to(dev)
    blk.compute_dtype = cd
    scale = max(1.0, ref.abs().max().item())
    try:
        N.check(L.llie_tune(b"se_mfma", 1))
        y1 = blk(x.to(dev), te.to(dev)).cpu()
        one = blk(x[:1].to(dev), te[:1].to(dev)).cpu()
        again = blk(x.to(dev), te.to(dev)).cpu()
        N.check(L.llie_tune(b"se_mfma", 0))
        y0 = blk(x.to(dev), te.to(dev)).cpu()
    finally:
        L.llie_tune(b"se_mfma", 1)
    assert torch.isfinite(y1).all()
    assert (y1 - ref).abs().max().item() < tol * scale, (y1 - ref).abs().max().item() / scale
    assert (y1 - y0).abs().max().item() < tol * scale
    assert torch.equal(one[0], y1[0]) and torch.equal(again, y1)


# ------------------------------------------------------------------ K segments of 64 n + 32 channels in the tile GEMM
@pytest.mark.parametrize("dtype,tdt,ulp", [(1, torch.float16, 2.0 ** -10), (2, torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("segs,nout", [([384, 64, 32], 32), ([128, 32], 64), ([96, 64, 32], 128), ([32, 96], 64), ([96, 96, 160], 32), ([160], 128),
                                       ([96], 32)])
def test_pw_gemm_half_empty_segment_tails(dev, dtype, tdt, ulp, segs, nout):
    """1x1 conv over a virtual concat (efficient_unet.py:186,199,265-267) whose segments are not all multiples of 64 channels
    (the 96 -> 32 and 32 -> 64 blocks of `small`, `base`'s 48 / 96 / 144-channel shapes): 64-wide K chunks with half-empty segment
    tails against float64, and bit for bit against the 32-wide chunks they replace (knob "gemm_bk" = 32)."""
    L = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, P, K = 2, 256, sum(segs)
    g = torch.Generator().manual_seed(K * 7 + nout)
    xs = [(torch.randn(B * P, c, generator=g) * 1.5).to(tdt) for c in segs]
    w32 = torch.randn(nout, K, generator=g) / math.sqrt(K)
    sc = (torch.rand(B, K, generator=g) + 0.5) / 6
    bi = (torch.randn(B, K, generator=g) * 0.5 + 0.4) / 6
    xd = [x.to(dev) for x in xs]
    scd, bid, wt = sc.to(dev), bi.to(dev), w32.to(dev).to(tdt)
    arr = (N.GemmSeg * len(xs))()
    off = 0
    for i, x in enumerate(xd):
        arr[i] = N.GemmSeg(x.data_ptr(), x.shape[1], scd.data_ptr() + off * 4, bid.data_ptr() + off * 4, K, 3)
        off += x.shape[1]
    outs = []
    try:
        for bk in (0, 32):
            N.check(L.llie_tune(b"gemm_bk", bk))
            out = torch.full((B * P, nout), float("nan"), dtype=tdt, device=dev)
            stats = torch.full((B, P // 128, 2, nout), float("nan"), device=dev)
            N.check(L.llie_pw_gemm(dtype, arr, len(xs), wt.data_ptr(), None, None, out.data_ptr(), stats.data_ptr(), B * P, nout, P, st), "pw_gemm")
            torch.cuda.synchronize()
            outs.append((out.cpu(), stats.cpu()))
    finally:
        L.llie_tune(b"gemm_bk", 0)
    (out, stats), (out32, stats32) = outs
    assert torch.isfinite(out.float()).all() and torch.isfinite(stats).all()
    assert torch.equal(out, out32) and torch.equal(stats, stats32)
    x = torch.cat([v.float() for v in xs], 1).view(B, P, -1)
    a = (x.double() * sc[:, None, :].double() + bi[:, None, :].double()).clamp(0, 1).to(tdt).double()
    ref = 6.0 * (a @ wt.cpu().double().t()).view(B * P, nout)
    err = (out.double() - ref).abs()
    tol = ulp * ref.abs() + ulp * 2.5
    assert (err <= tol).all(), (err / tol).max().item()


# ------------------------------------------------------------------ optimiser step: clip + AdamW + EMA in three launches
_OPT_SHAPES = [(3,), (32,), (5, 7), (128, 64, 1, 1), (4097,), (3, 32, 3, 3), (1,), (8192,), (300, 41), (512, 9)]


def _opt_case(dev, seed):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(dev)) for s in _OPT_SHAPES]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    return g, ps, qs


def _torch_reference_step(qs, ema, opt, max_norm, decay):
    norm = torch.nn.utils.clip_grad_norm_(qs, max_norm) if max_norm else None   # trainer.py:310-313
    opt.step()                                                                  # trainer.py:315
    for e, q in zip(ema, qs):                                                   # EMAModel.update, trainer.py:98-104
        e.mul_(decay).add_(q.data, alpha=1 - decay)
    return norm


@pytest.mark.parametrize("flat", [False, True])
def test_fused_adamw_matches_clip_grad_norm_adamw_and_ema(dev, flat):
    """FusedAdamW (llie_optimizer_step) against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW + the reference's EMA update
    over five steps: gradients in separate allocations (`step()`) and in one flat buffer with gaps (`step_flat`), the clip active
    in some steps and not in others.  fp32 on both sides, same operation order: a few units in the last place."""
    g, ps, qs = _opt_case(dev, 11)
    kw = dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    fused = M.FusedAdamW(ps, **kw, max_grad_norm=1.0, ema_decay=0.99)
    ref = torch.optim.AdamW(qs, **kw, foreach=False, fused=False)
    ema = [q.detach().clone() for q in qs]
    numel = [p.numel() for p in ps]
    offs, o = [], 5
    for n in numel:  # odd gaps: most gradients are not 16-byte aligned
        offs.append(o)
        o += n + 3
    for it in range(5):
        amp = [0.01, 3.0, 0.002, 10.0, 0.3][it]
        gs = [torch.randn(*s, generator=g).to(dev) * amp for s in _OPT_SHAPES]
        for q, gr in zip(qs, gs):
            q.grad = gr.clone()
        norm_ref = _torch_reference_step(qs, ema, ref, 1.0, 0.99)
        if flat:
            buf = torch.full((o,), float("nan"), device=dev)
            for gr, off, n in zip(gs, offs, numel):
                buf[off:off + n] = gr.reshape(-1)
            norm = fused.step_flat(buf, offs)
        else:
            for p, gr in zip(ps, gs):
                p.grad = gr.clone()
            fused.step()
            norm = fused.grad_norm()
        assert abs(norm.item() - norm_ref.item()) <= 2e-6 * norm_ref.item(), (it, norm.item(), norm_ref.item())
        for i, (p, q) in enumerate(zip(ps, qs)):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (it, i, (p - q).abs().max().item())
            st, sr = fused.state[p], ref.state[q]
            assert torch.allclose(st["exp_avg"], sr["exp_avg"], rtol=2e-6, atol=1e-9), (it, i)
            assert torch.allclose(st["exp_avg_sq"], sr["exp_avg_sq"], rtol=2e-6, atol=1e-12), (it, i)
        for e, er in zip(fused.ema_tensors(), ema):
            assert torch.allclose(e, er, rtol=2e-6, atol=1e-7), it
    assert not fused.last_step_skipped()


def test_fused_adamw_skips_a_nonfinite_step_and_keeps_torch_checkpoint_layout(dev):
    """skip_nonfinite (GradScaler.step's rule): an inf gradient leaves parameters, moments and shadows untouched and does not
    count as a step for nothing else either way; state_dict() loads into torch.optim.AdamW (the trainer's checkpoint layout,
    trainer.py:418-434) and back, and both continue identically; an LR scheduler drives param_groups[0]['lr']."""
    g, ps, qs = _opt_case(dev, 5)
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    fused = M.FusedAdamW(ps, **kw, max_grad_norm=0.5, ema_decay=0.9, skip_nonfinite=True)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(fused, T_max=10, eta_min=1e-4)
    ref = torch.optim.AdamW(qs, **kw, foreach=False, fused=False)
    sched_ref = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=10, eta_min=1e-4)
    for it in range(3):
        for p, q in zip(ps, qs):
            gr = torch.randn(p.shape, generator=g).to(dev)
            p.grad, q.grad = gr.clone(), gr.clone()
        torch.nn.utils.clip_grad_norm_(qs, 0.5)
        ref.step(); sched_ref.step()
        fused.step(); sched.step()
    assert abs(fused.param_groups[0]["lr"] - ref.param_groups[0]["lr"]) < 1e-12
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)
    before = [p.detach().clone() for p in ps] + [fused._m.clone(), fused._v.clone(), fused._ema.clone()]
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).to(dev)
    ps[3].grad[0, 0, 0, 0] = float("inf")
    fused.step()
    assert fused.last_step_skipped() and not math.isfinite(fused.grad_norm().item())
    after = [p.detach().clone() for p in ps] + [fused._m, fused._v, fused._ema]
    assert all(torch.equal(a, b) for a, b in zip(before, after))
    # checkpoint: ours -> torch.optim.AdamW -> one more step on both
    import copy
    sd = copy.deepcopy(fused.state_dict())  # as torch.save / torch.load would (state_dict() hands out references, like torch's)
    ema_flat = sd.pop("ema_shadow_flat")
    assert ema_flat.numel() == sum(p.numel() for p in ps)
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    other = torch.optim.AdamW(rs, **kw, foreach=False, fused=False)
    for st in sd["state"].values():  # the skipped update is no step of torch's either
        st["step"] = torch.tensor(3.0)
    other.load_state_dict(sd)
    fused._step = 3
    fused.skip_nonfinite = False
    other.param_groups[0]["lr"] = fused.param_groups[0]["lr"]
    for p, r in zip(ps, rs):
        gr = torch.randn(p.shape, generator=g).to(dev) * 0.1
        p.grad, r.grad = gr.clone(), gr.clone()
    torch.nn.utils.clip_grad_norm_(rs, 0.5)
    other.step()
    fused.step()
    for p, r in zip(ps, rs):
        assert torch.allclose(p, r, rtol=2e-6, atol=1e-7)
    # ... and torch's state_dict back into a fresh FusedAdamW
    again = M.FusedAdamW([torch.nn.Parameter(p.detach().clone()) for p in ps], **kw)
    again.load_state_dict(copy.deepcopy(other.state_dict()))
    assert again._step == 4
    assert torch.allclose(again._m, fused._m, rtol=2e-6, atol=1e-9) and torch.allclose(again._v, fused._v, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("cd", ["bf16", "fp32"])
def test_train_step_matches_the_autograd_path(dev, cd):
    """TrainStep (flat gradients, FusedAdamW) against compute_loss -> loss.backward() -> clip_grad_norm_ -> torch.optim.AdamW ->
    EMA on the engine's autograd node (trainer.py:281-324), small@64, B=2, same random draws: the same loss and the same
    gradients bit for bit, parameters and shadows equal to optimiser rounding after the step.  (Only one step is compared: at
    initialisation a 3e-7 difference in the parameters moves the next step's gradient norm by 6e-5 in fp32 and by 10 % in bf16.)
    Then two more steps, after which the engine must be running on the updated parameters: its output equals that of a fresh
    model loaded from state_dict() -- the optimiser writes the fp32 masters behind PyTorch's version counters."""
    import copy
    sched = M.LCMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear", prediction_type="v_prediction", rescale_betas_zero_snr=True)
    torch.manual_seed(3)
    a = M.LowLightDiffusion(unet_variant="small", image_size=64, compute_dtype=cd, scheduler=sched).to(dev).train()
    b = copy.deepcopy(a)
    low = synth_input("r4:tlow", (2, 3, 64, 64), -1.0, -0.2).to(dev)
    normal = synth_input("r4:tnormal", (2, 3, 64, 64), -1.0, 1.0).to(dev)
    pa, pb = list(a.parameters()), list(b.parameters())
    kw = dict(lr=1e-3, weight_decay=0.01)
    opt_a = torch.optim.AdamW(pa, **kw, foreach=False, fused=False)
    ema_a = [p.detach().clone() for p in pa]
    opt_b = M.FusedAdamW(pb, **kw, max_grad_norm=1.0, ema_decay=0.999)
    step_b = M.TrainStep(b, opt_b, loss_type="mse", use_velocity_target=True)
    torch.manual_seed(100)
    la = a.compute_loss(low, normal, loss_type="mse", use_velocity_target=True)
    la.backward()
    ga = [p.grad.detach().clone() for p in pa]
    norm_a = _torch_reference_step(pa, ema_a, opt_a, 1.0, 0.999)
    torch.manual_seed(100)
    lb = step_b(low, normal)
    assert torch.equal(la.detach(), lb), (la.item(), lb.item())
    for g_ref, off, p in zip(ga, step_b._offsets, pb):
        assert torch.equal(step_b._flat[off:off + p.numel()].view_as(p), g_ref)
    assert abs(opt_b.grad_norm().item() - norm_a.item()) <= 2e-6 * norm_a.item()
    for x, y in zip(pb, pa):
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-9), (x - y).abs().max().item()
    for e, er in zip(opt_b.ema_tensors(), ema_a):
        assert torch.allclose(e, er, rtol=2e-6, atol=2e-9)
    losses = [step_b(low, normal).item() for _ in range(2)]
    assert all(math.isfinite(v) for v in losses)
    fresh = M.LowLightDiffusion(unet_variant="small", image_size=64, compute_dtype=cd, scheduler=sched)
    fresh.load_state_dict(b.state_dict())
    fresh = fresh.to(dev).eval()
    b.eval()
    t = torch.tensor([500, 20], device=dev)
    with torch.no_grad():
        x = torch.cat([normal, low], 1)
        assert torch.equal(b.unet(x, t), fresh.unet(x, t))
    with pytest.raises(ValueError):
        M.TrainStep(b, M.FusedAdamW(pb[:-1], **kw))

