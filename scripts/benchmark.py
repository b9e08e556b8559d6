#!/usr/bin/env python3
"""Latency harness on the MI355X engine -- flag-compatible counterpart of the reference's
scripts/benchmark.py (:23-44): --model --format --image_size --batch_size --num_runs --warmup --num_steps
--device --threads, same timing protocol (:63-79: warm-up, then perf_counter around enhance +
cuda.synchronize) and report fields (:142-148).  `throughput_fps` is calls/s like the reference;
images/s (= calls/s x batch) is printed next to it.  Extensions: --variant, --dtype; --model may be
omitted (random-init weights).  The repository's headline benchmark is ../bench.py.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
hostio = importlib.import_module("cv-diffusion-model_amd.hostio")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Benchmark Model (HIP engine)")
    p.add_argument("--model", type=str, default=None, help="Model path (bare state_dict or trainer checkpoint)")
    p.add_argument("--format", type=str, default="pytorch", choices=["onnx", "tflite", "pytorch"])
    p.add_argument("--image_size", type=int, default=256)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--num_runs", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--num_steps", type=int, default=4)
    p.add_argument("--device", type=str, default="cuda", choices=["cpu", "cuda"])
    p.add_argument("--threads", type=int, default=4, help="(reference flag; unused: there is no CPU path)")
    p.add_argument("--variant", type=str, default="small", help="extension (the reference hard-codes small, benchmark.py:52)")
    p.add_argument("--dtype", type=str, default="fp32", choices=["fp32", "fp16", "bf16"], help="engine precision (extension)")
    return p.parse_args(argv)


def benchmark_pytorch(args):
    model = M.LowLightDiffusion(unet_variant=args.variant, image_size=args.image_size,
                                num_inference_steps=args.num_steps, compute_dtype=args.dtype)
    if args.model:
        hostio.load_checkpoint(model, args.model)
    model = model.to(args.device).eval()
    x = torch.randn(args.batch_size, 3, args.image_size, args.image_size).to(args.device)
    with torch.no_grad():
        for _ in range(args.warmup):
            model.enhance(x, num_inference_steps=args.num_steps)
    torch.cuda.synchronize()
    times = []
    with torch.no_grad():
        for _ in range(args.num_runs):
            start = time.perf_counter()
            model.enhance(x, num_inference_steps=args.num_steps)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - start)
    return times


def main(argv=None):
    args = parse_args(argv)
    if args.format != "pytorch":
        raise SystemExit(f"--format {args.format}: exported-model runtimes are outside this engine's scope; use --format pytorch")
    if args.device != "cuda":
        raise SystemExit("the engine has no CPU path; use --device cuda (for a CPU number see bench.py's cpu_baseline)")
    print("=" * 60)
    print("Low-Light Enhancement Model Benchmark")
    print("=" * 60)
    print("\nConfiguration:")
    for k in ("model", "format", "image_size", "batch_size", "device", "threads", "num_runs", "variant", "dtype"):
        print(f"  {k}: {getattr(args, k)}")
    print("\nRunning benchmark...")
    times = benchmark_pytorch(args)
    res = {"mean_latency_ms": np.mean(times) * 1000, "std_latency_ms": np.std(times) * 1000,
           "min_latency_ms": np.min(times) * 1000, "max_latency_ms": np.max(times) * 1000,
           "throughput_fps": 1.0 / np.mean(times)}
    print("\n" + "=" * 40 + "\nRESULTS\n" + "=" * 40)
    print(f"Mean latency:  {res['mean_latency_ms']:.2f} ms")
    print(f"Std latency:   {res['std_latency_ms']:.2f} ms")
    print(f"Min latency:   {res['min_latency_ms']:.2f} ms")
    print(f"Max latency:   {res['max_latency_ms']:.2f} ms")
    print(f"Throughput:    {res['throughput_fps']:.1f} FPS (calls/s) = {res['throughput_fps'] * args.batch_size:.1f} images/s")
    print(f"\nPer-step latency: {res['mean_latency_ms'] / args.num_steps:.2f} ms ({args.num_steps} steps)")
    target = 1000 / 30
    print(f"\nTarget: 30 FPS ({target:.1f} ms): " + ("met" if res["mean_latency_ms"] < target else
                                                   f"not met (needs {res['mean_latency_ms'] / target:.1f}x)"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
