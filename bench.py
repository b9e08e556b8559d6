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


def cpu_baseline(variant: str, size: int, steps: int, state_dict, budget_s: float = 20.0) -> dict:
    """The CPU oracle (a restatement of the reference's PyTorch path, pinned to reference goldens)
    timed on this box's host cores on a bounded sample of the same workload: enhance(B=1) repeated for
    about budget_s seconds (first call = warm-up unless it alone exceeds the budget)."""
    import oracle
    ncores = host_cores()
    torch.set_num_threads(ncores)
    spec = oracle.make_spec(variant, size, allow_unpinned=variant in ("tiny", "base"))
    sd = {k: v.detach().cpu().float() for k, v in state_dict.items()}
    g = torch.Generator().manual_seed(1234)
    low = torch.rand(1, 3, size, size, generator=g) * 2 - 1
    noise = oracle.draw_noise(1, size, steps, seed=123)
    t0 = time.perf_counter()
    oracle.enhance_ref(sd, spec, low, steps, noise)
    first = time.perf_counter() - t0
    log(f"cpu_baseline: first enhance(B=1) took {first:.1f} s on {ncores} threads")
    if first > budget_s / 2:
        n, el, note = 1, first, "1 cold call (no warm-up: a single call exceeds half the time budget)"
    else:
        n, t0 = 0, time.perf_counter()
        while True:
            oracle.enhance_ref(sd, spec, low, steps, noise)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 20:
                break
        note = f"{n} calls after 1 warm-up"
    return {"value": round(n / el, 4), "unit": "images/sec", "cores": ncores, "kind": "port",
            "sample": f"enhance(B=1, {variant}@{size}, {steps} steps, fp32), {note}, torch CPU threads={ncores}"}


def train_bench(args, M, dev, rank: int, world: int) -> None:
    """One step = the reference trainer's inner loop (trainer.py:281-338) on `--batch` images per GPU: q-sample,
    engine forward with kept activations, MSE, engine backward, one flat gradient all-reduce over the ranks,
    clip_grad_norm_(1.0), AdamW, EMA(0.9999).  Synthetic image pairs resident in HBM."""
    model = M.LowLightDiffusion(unet_variant=args.variant, image_size=args.image_size, compute_dtype=args.dtype).to(dev).train()
    B, S = args.batch, args.image_size
    g = torch.Generator().manual_seed(1234 + rank)
    low = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
    normal = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
    params = list(model.parameters())
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
    ema = [p.detach().clone() for p in params]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.compute_loss(low, normal, loss_type="mse")
        loss.backward()
        M.all_reduce_gradients(params)
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        torch._foreach_mul_(ema, 0.9999)
        torch._foreach_add_(ema, [p.detach() for p in params], alpha=1 - 0.9999)
        return loss

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
        print(json.dumps({
            "metric": "training images/sec (whole node), 256x256 'small' (BASELINE config 5)",
            "value": round(world * B * args.steps / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init weights, uniform image pairs)",
            "config": {"workload": f"variant={args.variant}, {S}x{S}, batch={B}/GPU, training step (MSE, AdamW, clip 1.0, EMA), "
                                   f"{args.dtype}, {world}xMI355X", "global_batch": world * B,
                       "parallelism": "single GPU" if world == 1 else f"dp{world} (one flat gradient all-reduce)"},
            "final_loss": round(float(loss.item()), 5)}), flush=True)


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
    breakdown = None
    for i in range(args.warmup):
        if prof and i == args.warmup - 1:       # last warm-up step: per-kernel breakdown of every profiled class
            handle.profile_begin(native.K_DW | native.K_GEMM | native.K_CONV3 | native.K_SE)
        step()
    drain()
    torch.cuda.synchronize()
    dom_class = native.K_DW
    if prof and args.warmup > 0:
        breakdown = handle.profile_report()
        dom_name = max(breakdown, key=lambda k: breakdown[k][0])
        dom_class = {"dwconv3x3": native.K_DW, "pw_gemm": native.K_GEMM, "conv3x3_k": native.K_CONV3}.get(dom_name[:9], native.K_SE)
    log("warm-up done; timing")
    if prof:
        handle.profile_begin(dom_class)   # HIP events around the dominant kernel's launches, inside the timed region
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
    line = {
        "metric": "images/sec (whole node), 256x256 4-step LCM 'small'",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init weights, uniform low-light batch, device noise)",
        "config": {"workload": f"variant={args.variant}, {S}x{S}, batch={B}/GPU, {args.lcm_steps} LCM steps, {args.dtype}, "
                               f"{world}xMI355X (BASELINE configs[1] per GPU)",
                   "global_batch": world * B, "parallelism": f"batch-shard x{world} + 1 all_gather" if world > 1 else "single GPU"},
    }
    if rank == 0:
        if not args.no_roofline:
            line["roofline"] = roofline(handle, native, args, breakdown)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.variant, S, args.lcm_steps, model.state_dict())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


def roofline(handle, native, args, breakdown=None) -> dict:
    """Roofline of the dominant kernel, from HIP events recorded on the launch stream during the timed
    region (engine hooks llie_profile_begin / llie_profile_report, aggregated per kernel name -- the
    granularity of `rocprofv3 --kernel-trace --stats`).  `achieved` = algorithmic bytes of that
    kernel's recorded launches / their summed device time (byte model: DESIGN.md section 3;
    depthwise: read h1 + write h2 = 2*B*P*Chid*elem; pointwise GEMM: A + out (+residual) + W).
    `traffic` = measured HBM bytes per launch from the committed PMC passes of this same command
    (profiles/r01/pmc_traffic.json; FETCH_SIZE x2 + WRITE_SIZE, see tools/pmc_summary.py), else null."""
    rep = handle.profile_report()
    if not rep:
        return None
    dom = max(rep, key=lambda k: rep[k][0])
    ms, n, nbytes = rep[dom]
    achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    default_cfg = (args.variant, args.image_size, args.batch, args.lcm_steps, args.dtype) == ("small", 256, 32, 4, "fp16")
    if default_cfg and os.path.exists(pmc):
        k = json.load(open(pmc)).get("kernels", {}).get(dom)
        if k:
            traffic = k["hbm_bytes_per_launch"]
    out = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
           "launches": n, "avg_launch_us": round(1e3 * ms / max(n, 1), 2), "alg_bytes_per_launch": int(nbytes / max(n, 1)),
           "forward_alg_bytes": handle.algorithmic_bytes(args.batch)}
    if breakdown:  # one warm-up step with every profiled class armed (not part of the timed region)
        tot = sum(v[0] for v in breakdown.values())
        out["warmup_step_breakdown"] = {k: {"ms": round(v[0], 3), "launches": v[1], "avg_us": round(1e3 * v[0] / v[1], 2),
                                            "GBps": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None,
                                            "share": round(v[0] / tot, 3)}
                                        for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][0])}
    return out


if __name__ == "__main__":
    main()
