#!/usr/bin/env python3
"""Golden alpha-bar tables of every beta schedule the reference's LCMScheduler offers (lcm_scheduler.py:77-88 `linear`,
`scaled_linear`, `squaredcos_cap_v2` = `_cosine_beta_schedule` :107-114), with and without the zero-SNR rescale (:116-129),
plus one `step` and one `add_noise` per schedule.  Runs only where /root/reference exists (the reference is imported with
the 3-symbol `diffusers` placeholder of tools/make_golden.py); writes tests/golden/schedules_kat.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import OUT, load_ref_models_package, np32, synth_input  # noqa: E402


def main():
    M = load_ref_models_package()
    out = {}
    sample = synth_input("sched2.sample", (2, 3, 8, 8), -3, 3)
    mo = synth_input("sched2.model_output", (2, 3, 8, 8), -2, 2)
    x0 = synth_input("sched2.x0", (3, 3, 8, 8))
    nz = synth_input("sched2.noise", (3, 3, 8, 8), -2, 2)
    tt = torch.tensor([3, 499, 998])
    for sched in ("linear", "scaled_linear", "squaredcos_cap_v2"):
        for rescale in (False, True):
            s = M.LCMScheduler(beta_schedule=sched, rescale_betas_zero_snr=rescale)
            tag = f"{sched}_{int(rescale)}"
            out[f"acp_{tag}"] = np32(s.alphas_cumprod)
            s.set_timesteps(4)
            t = int(s.timesteps[1])
            torch.manual_seed(77)
            r = s.step(mo, t, sample)
            out[f"step_{tag}_t"] = np.int64(t)
            out[f"step_{tag}_prev"] = np32(r.prev_sample)
            out[f"step_{tag}_x0"] = np32(r.pred_original_sample)
            out[f"add_noise_{tag}"] = np32(s.add_noise(x0, nz, tt))
    np.savez_compressed(os.path.join(OUT, "schedules_kat.npz"), **out)
    print("wrote schedules_kat.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
