#!/usr/bin/env python3
"""Headline benchmark: images/sec of the 4-step LCM `enhance` loop, variant small, 256x256, fp16,
batch 32 per GPU (BASELINE.json configs[1]), synthetic inputs and random-init weights.

    python bench.py --gpus N --steps K --warmup W

A "step" is one `LowLightDiffusion.enhance(x, 4)` call on this rank's batch (4 UNet forwards + 4
scheduler steps), inputs already resident in HBM.  For N > 1 the driver launches one process per GPU
with torch.distributed.run; ranks hold independent batch shards (weak scaling) and the outputs are
collected with a single RCCL all_gather inside the timed region.  Rank 0 prints one
JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def log(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota and by the
    GPU box's per-GPU CPU share (16)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, 16))


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:  # noqa: BLE001
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(variant: str, size: int, steps: int, state_dict, batch: int, budget_s: float = 24.0, train: bool = False) -> dict:
    """The CPU oracle (a restatement of the reference's PyTorch path, pinned to reference goldens)
    timed on this box's host cores on a bounded sample of the same workload (B=1 and B=min(B,4)):
    enhance(B=1) repeated for about 60 % of budget_s (first call = warm-up unless it alone exceeds the budget), then
    enhance(B=min(batch,4)) for the rest (at least one call).  `value` = the better of the two legs."""
    import oracle
    if train:
        # the training step on the CPU oracle (B=1): q-sample, oracle forward, v-pred MSE, PyTorch autograd, clip, AdamW
        import torch.nn.functional as F
        from oracle import scheduler_ref as S
        ncores = host_cores()
        torch.set_num_threads(ncores)
        spec = oracle.make_spec(variant, size)
        params = {k: v.detach().cpu().float().clone().requires_grad_(True) for k, v in state_dict.items()}
        opt = torch.optim.AdamW(list(params.values()), lr=1e-4, weight_decay=0.01)
        tab = S.LCMTables.build(rescale_betas_zero_snr=True)
        g = torch.Generator().manual_seed(1234)
        low, normal = torch.rand(1, 3, size, size, generator=g) * 2 - 1, torch.rand(1, 3, size, size, generator=g) * 2 - 1

        def step():
            t = torch.randint(0, 1000, (1,), generator=g)
            noise = torch.randn(1, 3, size, size, generator=g)
            opt.zero_grad()
            pred = oracle.unet_forward(params, spec, torch.cat([S.add_noise(tab, normal, noise, t), low], 1), t)
            F.mse_loss(pred, S.get_velocity(tab, normal, noise, t)).backward()
            torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
            opt.step()

        t0 = time.perf_counter()
        step()
        first = time.perf_counter() - t0
        log(f"cpu_baseline (training): first step took {first:.1f} s on {ncores} threads")
        n, el = 1, first
        if first < budget_s / 2:
            n, t0 = 0, time.perf_counter()
            while True:
                step()
                n += 1
                el = time.perf_counter() - t0
                if el > budget_s or n >= 10:
                    break
        return {"value": round(n / el, 4), "unit": "images/sec", "cores": ncores, "kind": "port",
                "sample": f"training step on the CPU oracle (B=1, {variant}@{size}, v-pred MSE, autograd, clip, AdamW), {n} step(s), "
                          f"torch CPU threads={ncores}"}
    ncores = host_cores()
    torch.set_num_threads(ncores)
    spec = oracle.make_spec(variant, size, allow_unpinned=variant in ("tiny", "base"))
    sd = {k: v.detach().cpu().float() for k, v in state_dict.items()}
    legs, kept = [], None
    for bb, share in ((1, 0.6), (min(batch, 4), 0.4)):
        if bb == 1 and legs:
            break  # batch == 1: one leg
        g = torch.Generator().manual_seed(1234)
        low = torch.rand(bb, 3, size, size, generator=g) * 2 - 1
        noise = oracle.draw_noise(bb, size, steps, seed=123)
        leg_budget = budget_s * share
        t0 = time.perf_counter()
        ref_out = oracle.enhance_ref(sd, spec, low, steps, noise)
        first = time.perf_counter() - t0
        if bb == 1:  # kept for the quality fields: the GPU engines are run on exactly these inputs (quality())
            kept = {"low": low, "noise": noise, "enhanced": ref_out["enhanced"]}
        log(f"cpu_baseline: first enhance(B={bb}) took {first:.1f} s on {ncores} threads")
        if first > leg_budget / 2:
            n, el, note = 1, first, "1 cold call (a single call exceeds half this leg's time budget)"
        else:
            n, t0 = 0, time.perf_counter()
            while True:
                oracle.enhance_ref(sd, spec, low, steps, noise)
                n += 1
                el = time.perf_counter() - t0
                if el > leg_budget or n >= 20:
                    break
            note = f"{n} calls after 1 warm-up"
        legs.append({"batch": bb, "images_per_sec": round(bb * n / el, 4), "note": note})
    best = max(legs, key=lambda l: l["images_per_sec"])
    return {"value": best["images_per_sec"], "unit": "images/sec", "cores": ncores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"enhance({variant}@{size}, {steps} steps, fp32) on torch CPU threads={ncores}: " +
                      "; ".join(f"B={l['batch']}: {l['images_per_sec']} img/s ({l['note']})" for l in legs),
            "legs": legs, "_kept": kept}


def quality(M, model, kept, args, dev) -> dict:
    """The second half of BASELINE's metric ("PSNR vs CPU ref"): the engine under test and the fp32 parity engine on the
    SAME image, weights and CPU-drawn noise as the CPU oracle's B=1 leg.  PSNR on [0, 1]-denormalised images ((x + 1) / 2,
    low_light_diffusion.py:417-419; MAX = 1); max-abs on the clamped [-1, 1] outputs (north_star's bar for fp32: 1e-3)."""
    import math
    low, noise, ref = kept["low"].to(dev), torch.stack(kept["noise"]).to(dev), kept["enhanced"].double()

    def psnr(x):
        mse = ((((x.double().clamp(-1, 1) + 1) / 2) - ((ref.clamp(-1, 1) + 1) / 2)) ** 2).mean().item()
        return 99.0 if mse == 0 else round(10 * math.log10(1.0 / mse), 2)
    out = model.enhance(low, args.lcm_steps, noise=noise).cpu()
    q = {"psnr_db_vs_cpu_ref": psnr(out), "max_abs_vs_cpu_ref": float(f"{(out.double() - ref).abs().max().item():.3e}"),
         "dtype": args.dtype, "sample": "B=1, the CPU baseline leg's image, weights and CPU-drawn noise (seed 123)"}
    if args.dtype != "fp32":
        m32 = M.LowLightDiffusion(unet_variant=args.variant, image_size=args.image_size, num_inference_steps=args.lcm_steps,
                                  compute_dtype="fp32", allow_unpinned_groupnorm=args.variant in ("tiny", "base"))
        m32.load_state_dict(model.state_dict())
        o32 = m32.to(dev).eval().enhance(low, args.lcm_steps, noise=noise).cpu()
        q["max_abs_fp32"] = float(f"{(o32.double() - ref).abs().max().item():.3e}")
        q["psnr_db_fp32"] = psnr(o32)
        q["fp32_bar"] = 1e-3
    else:
        q["max_abs_fp32"] = q["max_abs_vs_cpu_ref"]
    return q


def copy_probe(native, dev, mib: int = 1024, reps: int = 10) -> float:
    """Measured HBM copy bandwidth of this box in GB/s (llie_copy_probe: 16 bytes per lane, read + write counted),
    the `peak_measured` the roofline fractions are also divided by."""
    n = mib << 20
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream(dev)
    L = native.lib()
    for _ in range(2):
        native.check(L.llie_copy_probe(src.data_ptr(), dst.data_ptr(), n, st.cuda_stream), "copy_probe")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        native.check(L.llie_copy_probe(src.data_ptr(), dst.data_ptr(), n, st.cuda_stream), "copy_probe")
    e1.record(st)
    e1.synchronize()
    assert torch.equal(src[:4096], dst[:4096])
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def train_bench(args, M, dev, rank: int, world: int) -> None:
    """One step = the reference trainer's inner loop (trainer.py:281-338) on `--batch` images per GPU: q-sample,
    engine forward with kept activations, v-prediction MSE (BASELINE config 5: the velocity target of
    lcm_scheduler.py:282-305), engine backward, one flat gradient all-reduce over the ranks, clip_grad_norm_(1.0), AdamW,
    EMA(0.9999).  Synthetic image pairs resident in HBM."""
    sched = M.LCMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear", prediction_type="v_prediction",
                           rescale_betas_zero_snr=True)
    model = M.LowLightDiffusion(unet_variant=args.variant, image_size=args.image_size, compute_dtype=args.dtype,
                                scheduler=sched).to(dev).train()
    B, S = args.batch, args.image_size
    g = torch.Generator().manual_seed(1234 + rank)
    low = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
    normal = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
    params = list(model.parameters())
    if args.train_autograd:
        # the reference trainer's own calls on top of the engine's autograd node (any torch optimiser works this way)
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
        ema = [p.detach().clone() for p in params]

        def step():
            opt.zero_grad(set_to_none=True)
            loss = model.compute_loss(low, normal, loss_type="mse", use_velocity_target=True)
            loss.backward()
            M.all_reduce_gradients(params)
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            torch._foreach_mul_(ema, 0.9999)
            torch._foreach_add_(ema, [p.detach() for p in params], alpha=1 - 0.9999)
            return loss
    else:
        # the same arithmetic without the autograd graph: flat gradients, one all-reduce, clip + AdamW + EMA in three launches
        opt = M.FusedAdamW(params, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.9999)
        train_step = M.TrainStep(model, opt, loss_type="mse", use_velocity_target=True)

        def step():
            return train_step(low, normal)

    log(f"training model built on {dev}; warm-up x{args.warmup}")
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        native = importlib.import_module("cv-diffusion-model_amd._native")
        extra = {}
        if not args.no_roofline:
            # byte model of one training step: the forward's SURVEY.md 8d bytes (every activation is
            # written once and read once) plus a backward pass that reads each kept activation and each incoming gradient
            # and writes each outgoing gradient: 3 x llie_algorithmic_bytes.  The whole step (optimiser included) is timed.
            handle = model.unet._prepare(1, dev)[0]
            fwd = handle.algorithmic_bytes(B)
            peak_measured = copy_probe(native, dev)
            ach = 3 * fwd / (elapsed / args.steps) / 1e9
            extra["roofline"] = {"bound": "hbm", "kernel": "whole training step (forward + backward + optimiser)",
                                 "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                 "traffic": None, "peak_measured": round(peak_measured, 1),
                                 "frac_of_measured": round(ach / peak_measured, 4), "alg_bytes_per_step": 3 * fwd,
                                 "note": "3 x forward algorithmic bytes (SURVEY.md 8d model) / measured step time"}
        if not args.no_cpu_baseline and world == 1:
            extra["cpu_baseline"] = cpu_baseline(args.variant, S, 0, model.state_dict(), 1, budget_s=20.0, train=True)
        print(json.dumps({
            "metric": "training images/sec (whole node), 256\u00d7256 'small' (BASELINE config 5)",
            "value": round(world * B * args.steps / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init weights, uniform image pairs)",
            "config": {"workload": f"variant={args.variant}, {S}x{S}, batch={B}/GPU, training step (v-pred MSE, AdamW, clip 1.0, EMA), "
                                   f"{args.dtype}, {world}xMI355X", "global_batch": world * B,
                       "parallelism": "single GPU" if world == 1 else f"dp{world} (one flat gradient all-reduce)"},
            "final_loss": round(float(loss.item()), 5), **extra}), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--variant", default="small")
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--lcm_steps", type=int, default=4)
    ap.add_argument("--dtype", default="fp16", choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--train-autograd", dest="train_autograd", action="store_true",
                    help="with --train: loss.backward() + clip_grad_norm_ + torch.optim.AdamW + foreach EMA on the engine's autograd "
                         "node instead of TrainStep / FusedAdamW (same arithmetic)")
    ap.add_argument("--train", action="store_true",
                    help="time the training step instead (BASELINE config 5; not the headline line): compute_loss -> "
                         "backward -> gradient all-reduce -> clip -> AdamW -> EMA, as src/training/trainer.py:281-338")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # Rehearsal on a one-GPU box (the multi-rank code path otherwise only ever runs on the driver's 8-GPU node):
    # LLIE_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo (RCCL wants one device per rank).
    rehearsal = os.environ.get("LLIE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    M = importlib.import_module("cv-diffusion-model_amd")
    torch.manual_seed(0)
    if args.train:
        train_bench(args, M, dev, rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    model = M.LowLightDiffusion(unet_variant=args.variant, image_size=args.image_size,
                                num_inference_steps=args.lcm_steps, compute_dtype=args.dtype,
                                allow_unpinned_groupnorm=args.variant in ("tiny", "base"))  # opt-in, parity-unpinned
    model = model.to(dev).eval()
    B, S = args.batch, args.image_size
    g = torch.Generator().manual_seed(1234 + rank)
    low = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)      # synthetic low-light batch, resident in HBM
    # The path's only collective: all_gather of the outputs.  It is issued asynchronously (RCCL runs it on its own
    # stream) into one of two buffers, so step k's gather overlaps step k+1's compute; a buffer is reused only after
    # its previous gather has been waited for, and the last gather is waited for before the closing barrier.
    gather_bufs = [torch.empty(world * B, 3, S, S, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    nstep = [0]

    def step():
        out = model.enhance(low, args.lcm_steps)                     # noise drawn on device, reference order
        if world > 1:
            i = nstep[0] & 1
            if pending[i] is not None:
                pending[i].wait()
            pending[i] = dist.all_gather_into_tensor(gather_bufs[i], out, async_op=True)
            nstep[0] += 1
        return out

    def drain():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    log(f"model built on {dev}; warm-up x{args.warmup}")
    native = importlib.import_module("cv-diffusion-model_amd._native")
    handle = model.unet._prepare(B, dev)[0]
    prof = rank == 0 and not args.no_roofline
    # The timed region runs the SHIPPING path: profiling disarmed, so `enhance` replays its captured hipGraph (the
    # engine falls back to ~770 eager launches while per-kernel events are armed).  The per-kernel numbers behind
    # `roofline` come from a separate pass of the same steps right after the timed region (see roofline()).
    for i in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    log("warm-up done; timing")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    images = world * B * args.steps
    value = images / elapsed
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    breakdown, dom_prof = None, None
    if prof:
        # (rank 0 only: these passes call the model directly -- no collective, the other ranks are not here)
        # pass 1 (one step): every kernel class bracketed by HIP events on the launch stream -> which kernel dominates
        handle.profile_begin(ALL_CLASSES)
        model.enhance(low, args.lcm_steps)
        torch.cuda.synchronize()
        breakdown = handle.profile_report()
        dom_name = max(breakdown, key=lambda k: breakdown[k][0])
        dom_class = kernel_class_of(dom_name, native)
        # pass 2 (`--steps` steps): only the dominant kernel's launches bracketed, so that nothing else perturbs them
        handle.profile_begin(dom_class)
        for _ in range(args.steps):
            model.enhance(low, args.lcm_steps)
        torch.cuda.synchronize()
        dom_prof = handle.profile_report()
        dom_prof = {k: v for k, v in dom_prof.items() if k == dom_name} or dom_prof
    line = {
        "metric": baseline_metric(),
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init weights, uniform low-light batch, device noise)",
        "config": {"workload": f"variant={args.variant}, {S}x{S}, batch={B}/GPU, {args.lcm_steps} LCM steps, {args.dtype}, "
                               f"{world}xMI355X (BASELINE configs[1] per GPU)",
                   "global_batch": world * B, "parallelism": f"batch-shard x{world} + 1 all_gather" if world > 1 else "single GPU"},
    }
    if rank == 0:
        line["graph_replay"] = os.environ.get("LLIE_NO_GRAPH") is None  # the timed calls replayed the captured hipGraph
        if not args.no_roofline:
            peak_measured = copy_probe(native, dev)
            line["roofline"] = roofline(handle, native, args, breakdown, dom_prof, elapsed / args.steps, peak_measured)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.variant, S, args.lcm_steps, model.state_dict(), B)
            kept = line["cpu_baseline"].pop("_kept")
            line["quality"] = quality(M, model, kept, args, dev)  # "PSNR vs CPU ref": the metric's second half
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


ALL_CLASSES = 31


def baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim (the driver matches the bench line against it)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        return "images/sec (whole node), 256\u00d7256 4-step LCM 'small'; PSNR vs CPU ref"


def kernel_class_of(name: str, native) -> int:
    if name.startswith(("pw_gemm", "expand_stats")):
        return native.K_GEMM
    if name.startswith(("dwconv3x3", "expand_dw")):
        return native.K_DW
    if name.startswith("conv3x3"):
        return native.K_CONV3
    if name.startswith("se_"):
        return native.K_SE
    return native.K_OTHER


def kernel_family_of(name: str) -> str:
    """Kernel family of a launcher-recorded kernel name (roofline.families)."""
    if name.startswith(("pw_gemm", "pw_expand")):
        return "pointwise GEMM (pw_gemm + pw_expand)"
    if name.startswith("expand_dw"):
        return "expand_dw (recompute expand + depthwise)"
    if name.startswith(("expand_stats", "gram_stats")):
        return "norm2 statistics passes (gram_stats, expand_stats)"
    if name.startswith("dwconv3x3"):
        return "dwconv3x3"
    if name.startswith("conv3x3"):
        return "dense 3x3 convs (down / up-sampling)"
    if name.startswith(("init_conv", "final_conv")):
        return "input / output heads"
    if name.startswith(("gn_finalize", "gram_finalize", "se_", "time_embed", "film", "zero_fill")):
        return "small launches (finalize, SE, time / FiLM)"
    if name.startswith(("linattn", "affine_add")):
        return "linear attention"
    return "other"


def pmc_lookup(kernels: dict, name: str):
    """HBM bytes per launch of kernel `name` in a tools/pmc_summary.py table.  Its demangler prints template arguments it
    cannot resolve (bool / defaulted ones) as `?`: `pw_gemm_kernel<_Float16, 128, 128, 2, 2, 64, ?, ?, ?>` is the launcher's
    `pw_gemm_kernel<_Float16, 128, 128, 2, 2, 64>`; the longest prefix match on the resolved arguments wins."""
    best, best_len = None, -1
    for key, v in kernels.items():
        stem = key.replace(", ?", "").rstrip(">")
        if (name == stem + ">" or name.startswith(stem + ",")) and len(stem) > best_len:
            best, best_len = v, len(stem)
    return best["hbm_bytes_per_launch"] if best else None


def roofline(handle, native, args, breakdown, dom_prof, step_seconds: float, peak_measured: float) -> dict:
    """Roofline of the dominant kernel from HIP events recorded on the launch stream (engine hooks llie_profile_begin /
    llie_profile_report, aggregated per kernel name -- the granularity of `rocprofv3 --kernel-trace --stats`), collected
    in a pass of `--steps` steps right AFTER the timed region, because the timed region itself replays a hipGraph whose
    kernels cannot be bracketed individually.  `achieved` = algorithmic bytes of that kernel's recorded launches / their
    summed device time (byte model per kernel: DESIGN.md section 3).  `traffic` = measured HBM bytes per launch from the
    committed PMC passes of this same command (profiles/r04/pmc_traffic.json, else an earlier round's; FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_summary.py), else null.  `whole_path` is the same quotient for everything `enhance` does, measured over the
    timed region itself: `whole_path.frac` = lcm_steps x llie_path_bytes (the bytes the engine's own launch sequence has to
    move: recompute blocks charged 3Cin + 2Chid + Cout, since round 3) / the step time; the reference-shaped model of SURVEY.md
    8d (llie_algorithmic_bytes: h1 written and read, what round 2's `whole_path.frac` used) is `whole_path.materialised_model`.
    `families` = share of the step and achieved GB/s per kernel family (the per-instantiation split of `kernel` hides that the
    pointwise-GEMM family and expand_dw are larger than the dominant instantiation); `worst` = the kernel with at least
    5 % of the step that is furthest below the HBM roof."""
    if not dom_prof:
        return None
    dom = max(dom_prof, key=lambda k: dom_prof[k][0])
    ms, n, nbytes = dom_prof[dom]
    achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None
    pmc = next((q for q in (os.path.join(ROOT, "profiles", r, "pmc_traffic.json") for r in ("r04", "r03", "r02")) if os.path.exists(q)), "")
    default_cfg = (args.variant, args.image_size, args.batch, args.lcm_steps, args.dtype) == ("small", 256, 32, 4, "fp16")
    if default_cfg and os.path.exists(pmc):
        traffic = pmc_lookup(json.load(open(pmc)).get("kernels", {}), dom)
    fwd = handle.algorithmic_bytes(args.batch)
    path = handle.path_bytes(args.batch)
    whole = args.lcm_steps * fwd / step_seconds / 1e9
    out = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
           "peak_measured": round(peak_measured, 1), "frac_of_measured": round(achieved / peak_measured, 4),
           "launches": n, "avg_launch_us": round(1e3 * ms / max(n, 1), 2), "alg_bytes_per_launch": int(nbytes / max(n, 1)),
           "measured_in": f"{args.steps} eager steps right after the timed region (the timed steps replay a hipGraph)",
           "forward_alg_bytes": fwd,
           # the engine runs 10 of 22 blocks in the recompute form, so SURVEY.md 8d asks for the recompute byte model here:
           # `frac` is what the engine's own launch sequence has to move (llie_path_bytes) over the timed region; the
           # reference-shaped (materialised) model is the secondary figure
           "whole_path": {"model": "engine launch sequence: recompute blocks charged 3Cin+2Chid+Cout (llie_path_bytes)",
                          "alg_bytes_per_step": args.lcm_steps * path, "achieved": round(args.lcm_steps * path / step_seconds / 1e9, 1),
                          "frac": round(args.lcm_steps * path / step_seconds / 1e9 / HBM_PEAK_GBS, 4),
                          "frac_of_measured": round(args.lcm_steps * path / step_seconds / 1e9 / peak_measured, 4),
                          "materialised_model": {"note": "SURVEY.md 8d reference-shaped model (2Cin+4Chid+Cout per block: h1 written and read)",
                                                 "alg_bytes_per_step": args.lcm_steps * fwd, "achieved": round(whole, 1),
                                                 "frac": round(whole / HBM_PEAK_GBS, 4),
                                                 "frac_of_measured": round(whole / peak_measured, 4)}}}
    if breakdown:  # one post-timing step with every class armed
        tot = sum(v[0] for v in breakdown.values())
        fams = {}
        for k, v in breakdown.items():
            f = fams.setdefault(kernel_family_of(k), [0.0, 0, 0])
            f[0] += v[0]; f[1] += v[1]; f[2] += v[2]
        out["families"] = {k: {"ms": round(v[0], 3), "launches": v[1], "share": round(v[0] / tot, 3),
                               "GBps": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 and v[2] > 0 else None,
                               "frac": round(v[2] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[0] > 0 and v[2] > 0 else None}
                           for k, v in sorted(fams.items(), key=lambda kv: -kv[1][0])}
        big = {k: v for k, v in breakdown.items() if v[0] / tot >= 0.05 and v[2] > 0}
        if big:
            wk = min(big, key=lambda k: big[k][2] / big[k][0])
            wv = big[wk]
            out["worst"] = {"kernel": wk, "share": round(wv[0] / tot, 3), "avg_launch_us": round(1e3 * wv[0] / wv[1], 2),
                            "achieved": round(wv[2] / (wv[0] * 1e-3) / 1e9, 1),
                            "frac": round(wv[2] / (wv[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "kernel with >= 5 % of the step furthest below the 8 TB/s roof (algorithmic bytes / event time)"}
        out["step_breakdown"] = {k: {"ms": round(v[0], 3), "launches": v[1], "avg_us": round(1e3 * v[0] / v[1], 2),
                                     "GBps": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None,
                                     "share": round(v[0] / tot, 3)}
                                 for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][0])}
    return out


if __name__ == "__main__":
    main()
