"""fp32 CPU restatement of LowLightDiffusion.enhance (low_light_diffusion.py:177-248).  TEST INFRASTRUCTURE."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .scheduler_ref import LCMTables, lcm_step, lcm_timesteps
from .spec import UNetSpec
from .unet_ref import unet_forward


def draw_noise(batch: int, size: int, steps: int, seed: int) -> List[torch.Tensor]:
    """Noise in the reference's draw order on the CPU generator: one randn for the initial latents
    (low_light_diffusion.py:208-211) then one randn_like per step except the last
    (lcm_scheduler.py:228-237).  The reference uses the *global* RNG for the per-step draws; seeding
    the global generator and passing no `generator` reproduces exactly this sequence."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, 3, size, size, generator=g) for _ in range(steps)]


@torch.no_grad()
def enhance_ref(sd: Dict[str, torch.Tensor], spec: UNetSpec, low_light: torch.Tensor, num_inference_steps: int,
                noise: Sequence[torch.Tensor], tables: Optional[LCMTables] = None) -> dict:
    """Returns {"enhanced", "intermediate" (post-step pre-clamp latents), "noise_pred" (per step)}."""
    tab = tables or LCMTables.build()
    ts = lcm_timesteps(num_inference_steps, tab.num_train_timesteps, tab.original_inference_steps)
    b = low_light.shape[0]
    latents = noise[0]
    inter, preds = [], []
    for i, t in enumerate(ts):
        prev_t = ts[i + 1] if i + 1 < len(ts) else 0
        tt = torch.full((b,), t, dtype=torch.long)
        eps = unet_forward(sd, spec, torch.cat([latents, low_light], dim=1), tt)
        latents, _ = lcm_step(tab, eps, t, prev_t, latents, noise[i + 1] if prev_t != 0 else None)
        preds.append(eps)
        inter.append(latents)
    return dict(enhanced=latents.clamp(-1, 1), intermediate=inter, noise_pred=preds, timesteps=ts)
