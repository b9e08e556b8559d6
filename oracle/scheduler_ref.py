"""fp32 CPU restatement of LCMScheduler (lcm_scheduler.py:53-305).  TEST INFRASTRUCTURE.

No `diffusers` dependency: that package only contributes two empty mixins and a config decorator
to the reference (lcm_scheduler.py:23-24,53); all arithmetic is restated here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch


@dataclass
class LCMTables:
    """beta/alpha-bar tables, lcm_scheduler.py:77-100 (built in fp32 like the reference)."""
    alphas_cumprod: torch.Tensor
    final_alpha_cumprod: torch.Tensor
    num_train_timesteps: int = 1000
    original_inference_steps: int = 50
    prediction_type: str = "epsilon"

    @staticmethod
    def build(num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
              beta_schedule: str = "scaled_linear", rescale_betas_zero_snr: bool = True,
              prediction_type: str = "epsilon", original_inference_steps: int = 50) -> "LCMTables":
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps) ** 2
        elif beta_schedule == "squaredcos_cap_v2":  # _cosine_beta_schedule, lcm_scheduler.py:107-114 (s = 0.008)
            n, s = num_train_timesteps, 0.008
            x = torch.linspace(0, n, n + 1)
            bar = torch.cos(((x / n) + s) / (1 + s) * math.pi * 0.5) ** 2
            bar = bar / bar[0]
            betas = torch.clip(1 - (bar[1:] / bar[:-1]), 0, 0.999)
        else:
            raise ValueError(f"Unknown beta schedule: {beta_schedule}")
        acp = torch.cumprod(1.0 - betas, dim=0)
        if rescale_betas_zero_snr:  # lcm_scheduler.py:116-129; LowLightDiffusion turns it on (low_light_diffusion.py:102)
            s = acp.sqrt()
            s0, sT = s[0].clone(), s[-1].clone()
            s = (s - sT) * (s0 / (s0 - sT))
            acp = s ** 2
        return LCMTables(acp, acp[0], num_train_timesteps, original_inference_steps, prediction_type)


def lcm_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000,
                  original_inference_steps: int = 50) -> List[int]:
    """lcm_scheduler.py:150-161.  n=4 -> [739,499,259,19] (the docstring at :141 is wrong)."""
    c = num_train_timesteps // original_inference_steps
    origin = [(i + 1) * c - 1 for i in range(original_inference_steps)]
    skip = len(origin) // num_inference_steps
    if skip == 0:
        raise ValueError("slice step cannot be zero")
    return list(reversed(origin[::skip][:num_inference_steps]))


def lcm_step(tab: LCMTables, model_output: torch.Tensor, t: int, prev_t: int, sample: torch.Tensor,
             noise: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """LCMScheduler.step, lcm_scheduler.py:204-242.  prev_t is the next list entry, 0 after the last
    (:171-174).  Returns (prev_sample, pred_original_sample).  Scalars stay 0-d fp32 tensors so
    that rounding matches the reference's tensor**0.5 arithmetic."""
    a_t = tab.alphas_cumprod[t]
    a_p = tab.alphas_cumprod[prev_t] if prev_t > 0 else tab.final_alpha_cumprod
    b_t, b_p = 1 - a_t, 1 - a_p
    if tab.prediction_type == "epsilon":
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    elif tab.prediction_type == "v_prediction":
        x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
    else:
        raise ValueError(f"Unknown prediction type: {tab.prediction_type}")
    if prev_t == 0:
        return x0, x0
    return a_p ** 0.5 * x0 + b_p ** 0.5 * noise, x0


def _bcast(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    while v.dim() < like.dim():
        v = v.unsqueeze(-1)
    return v


def add_noise(tab: LCMTables, x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """lcm_scheduler.py:255-280."""
    a = tab.alphas_cumprod.to(x0.dtype)[t]
    return _bcast(a ** 0.5, x0) * x0 + _bcast((1 - a) ** 0.5, x0) * noise


def get_velocity(tab: LCMTables, sample: torch.Tensor, noise: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """lcm_scheduler.py:282-305."""
    a = tab.alphas_cumprod.to(sample.dtype)[t]
    return _bcast(a ** 0.5, sample) * noise - _bcast((1 - a) ** 0.5, sample) * sample


# ---------------------------------------------------------------------------------------------
# Deployment loop (src/export/android_pipeline.py:191-277): numpy float64 tables, no zero-SNR rescale,
# x0 clamped to [-1, 1] before re-noising.  Restated in float64 numpy like the reference computes it
# (np.float64 scalars promote the float32 arrays under NumPy >= 2).
def deploy_alphas_cumprod(num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012):
    import numpy as np
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps) ** 2  # :209
    return np.cumprod(1.0 - betas)                                                      # :210-211


def deploy_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000):
    import numpy as np
    c = num_train_timesteps // 50                          # :218-219
    lcm = np.arange(1, 51) * c - 1                         # :221
    skip = len(lcm) // num_inference_steps                 # :223
    return lcm[::skip][:num_inference_steps][::-1].copy()  # :224-226


def deploy_step(acp, timesteps, noise_pred, t: int, sample, noise):
    """LCMDenoisingLoop.step (:240-265); `noise` replaces the np.random.randn draw of :262."""
    import numpy as np
    idx = int(np.where(timesteps == t)[0][0])
    prev_t = int(timesteps[idx + 1]) if idx + 1 < len(timesteps) else 0
    a_t = acp[t]
    a_p = acp[prev_t] if prev_t > 0 else acp[0]
    x0 = np.clip((sample - np.sqrt(1 - a_t) * noise_pred) / np.sqrt(a_t), -1, 1)
    if prev_t == 0:
        return x0
    return np.sqrt(a_p) * x0 + np.sqrt(1 - a_p) * noise


def deploy_add_noise(acp, original, noise, t: int):
    import numpy as np
    return np.sqrt(acp[t]) * original + np.sqrt(1 - acp[t]) * noise  # :228-238
