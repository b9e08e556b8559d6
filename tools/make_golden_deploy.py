#!/usr/bin/env python3
"""Golden vectors of the deployment denoising loop (`LCMDenoisingLoop`, src/export/android_pipeline.py:191-277):
scaled-linear alpha-bar table WITHOUT the zero-SNR rescale, float64 numpy scalars, and an x0 clamp before
re-noising.  Runs only where /root/reference exists; writes tests/golden/deploy_loop_kat.npz.

The class is loaded from its file (the module imports only torch / numpy at import time).
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle.weightgen import uniform01  # noqa: E402


def synth(name, shape, lo, hi):
    n = int(np.prod(shape))
    return (uniform01(name, n).astype(np.float64) * (hi - lo) + lo).astype(np.float32).reshape(shape)


def main():
    spec = importlib.util.spec_from_file_location("_ref_android_pipeline", os.path.join(REF, "src/export/android_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    out = {}
    loop = mod.LCMDenoisingLoop(num_inference_steps=4)
    out["alphas_cumprod"] = np.asarray(loop.alphas_cumprod)  # float64
    for n in (4, 6, 8):
        out[f"timesteps_{n}"] = np.asarray(mod.LCMDenoisingLoop(num_inference_steps=n).timesteps, dtype=np.int64)
    sample = synth("deploy.sample", (2, 3, 8, 8), -3, 3)
    eps = synth("deploy.noise_pred", (2, 3, 8, 8), -2, 2)
    out["sample"], out["noise_pred"] = sample, eps
    for t in loop.timesteps.tolist():
        np.random.seed(2000 + t)
        res = loop.step(eps, t, sample)
        np.random.seed(2000 + t)
        out[f"noise_{t}"] = np.random.randn(*sample.shape).astype(np.float32)  # the draw step() made (:262)
        out[f"step_{t}"] = np.asarray(res)  # dtype as numpy's promotion rules made it
    x0 = synth("deploy.x0", (2, 3, 8, 8), -1, 1)
    nz = synth("deploy.add_noise", (2, 3, 8, 8), -2, 2)
    out["x0"], out["add_noise_noise"] = x0, nz
    for t in (19, 499, 999):
        out[f"add_noise_{t}"] = np.asarray(loop.add_noise(x0, nz, t))
    out["numpy_version"] = np.array(np.__version__)
    path = os.path.join(ROOT, "tests", "golden", "deploy_loop_kat.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), {k: (v.dtype, v.shape) for k, v in out.items() if k.startswith("step_")})


if __name__ == "__main__":
    main()
