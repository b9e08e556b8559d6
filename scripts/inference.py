#!/usr/bin/env python3
"""Image / folder enhancement CLI on the MI355X engine -- flag-compatible counterpart of the reference's
scripts/inference.py (:30-62): --input --output --checkpoint --model --format --variant --image_size
--num_steps --device.  Only --format pytorch exists here (ONNX / TFLite are the reference's mobile
deployment targets, out of scope); extension: --dtype {fp32,fp16,bf16}.
"""
import argparse
import importlib
import os
import sys
import time
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("cv-diffusion-model_amd")
hostio = importlib.import_module("cv-diffusion-model_amd.hostio")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Low-Light Enhancement Inference (HIP engine)")
    p.add_argument("--input", type=str, required=True, help="Input image or folder")
    p.add_argument("--output", type=str, required=True, help="Output image or folder")
    p.add_argument("--checkpoint", type=str, default=None, help="PyTorch checkpoint")
    p.add_argument("--model", type=str, default=None, help="(reference flag; exported ONNX/TFLite models are not supported)")
    p.add_argument("--format", type=str, default="pytorch", choices=["pytorch", "onnx", "tflite"])
    p.add_argument("--variant", type=str, default="small")
    p.add_argument("--image_size", type=int, default=256)
    p.add_argument("--num_steps", type=int, default=4)
    p.add_argument("--device", type=str, default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--dtype", type=str, default="fp32", choices=["fp32", "fp16", "bf16"], help="engine precision (extension)")
    p.add_argument("--noise_seed", type=int, default=None,
                   help="(extension) draw the loop's noise on the CPU generator with this seed, in the reference's order, "
                        "instead of on the device: makes a run reproducible against the CPU reference")
    return p.parse_args(argv)


def load_model(args):
    if args.format != "pytorch":
        raise SystemExit(f"--format {args.format}: exported-model runtimes are outside this engine's scope; use --format pytorch")
    model = M.LowLightDiffusion(unet_variant=args.variant, image_size=args.image_size, num_inference_steps=4,
                                compute_dtype=args.dtype)
    if args.checkpoint:
        hostio.load_checkpoint(model, args.checkpoint)
    return model.to(args.device).eval()


def process_single_image(args, model, input_path: str, output_path: str) -> float:
    print(f"Processing: {input_path}")
    rgb = hostio.load_image(input_path)
    original = rgb.shape[:2]
    start = time.perf_counter()
    with torch.no_grad():
        # uint8 goes up, uint8 comes back: resize + normalise / denormalise run on the device
        # (bit-exact twins of hostio.preprocess_array / postprocess_array, i.e. inference.py:99-134)
        x = hostio.preprocess_device(torch.from_numpy(rgb).to(args.device), args.image_size)
        noise = None
        if args.noise_seed is not None:  # reference draw order: initial latents, then one draw per non-final step
            g = torch.Generator().manual_seed(args.noise_seed)
            noise = torch.stack([torch.randn(1, 3, args.image_size, args.image_size, generator=g) for _ in range(args.num_steps)])
        enhanced = model.enhance(x, num_inference_steps=args.num_steps, noise=noise)
        out = hostio.postprocess_device(enhanced, original)[0].cpu().numpy()
    elapsed = time.perf_counter() - start
    hostio.save_image(output_path, out)
    print(f"  Saved to: {output_path}")
    print(f"  Time: {elapsed * 1000:.1f} ms")
    return elapsed


def main(argv=None):
    args = parse_args(argv)
    print("=" * 60)
    print("Low-Light Enhancement Inference")
    print("=" * 60)
    print(f"\nLoading model ({args.format})...")
    model = load_model(args)
    print("Model loaded!")
    inp, out = Path(args.input), Path(args.output)
    if inp.is_file():
        out.parent.mkdir(parents=True, exist_ok=True)
        process_single_image(args, model, str(inp), str(out))
    elif inp.is_dir():
        out.mkdir(parents=True, exist_ok=True)
        images = sorted(f for f in inp.iterdir() if f.suffix.lower() in {".jpg", ".jpeg", ".png", ".bmp"})
        print(f"\nProcessing {len(images)} images...")
        total = sum(process_single_image(args, model, str(f), str(out / f.name)) for f in images)
        if images:
            avg = total / len(images) * 1000
            print(f"\nAverage time per image: {avg:.1f} ms")
            print(f"Throughput: {1000 / avg:.1f} FPS")
    else:
        print(f"Error: {inp} not found")
        return 1
    print("\nDone!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
