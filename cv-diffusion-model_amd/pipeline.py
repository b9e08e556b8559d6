"""Drop-in `LowLightDiffusion` (src/models/low_light_diffusion.py:31-281) on the HIP engine.

Constructor, `forward` / `enhance` / `compute_loss` / `get_model_size` signatures, attributes
(`.unet`, `.scheduler`, `.image_size`, `.condition_mode`) and the `state_dict` layout (all keys under
`unet.`) are the reference's, so scripts/inference.py and scripts/benchmark.py of the reference can
call it unchanged.  The whole denoising loop runs as one launch sequence in libllie_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from .scheduler import LCMScheduler
from .unet import EfficientUNet, create_efficient_unet


@dataclass
class LowLightDiffusionOutput:
    enhanced: torch.Tensor
    intermediate: Optional[list] = None


class LowLightDiffusion(nn.Module):
    def __init__(self, unet: Optional[EfficientUNet] = None, scheduler: Optional[LCMScheduler] = None,
                 unet_variant: str = "small", image_size: int = 256, num_inference_steps: int = 4,
                 condition_mode: str = "concat", compute_dtype: Optional[str] = None,
                 allow_unpinned_groupnorm: bool = False):
        """Arguments as in low_light_diffusion.py:50-58.  `compute_dtype` (extension; "fp32" | "fp16" |
        "bf16") pins the engine precision; when None the engine runs fp32, or the dtype of an active
        `torch.autocast("cuda")` region."""
        super().__init__()
        if condition_mode != "concat":
            # "add" routes low_light through a small conv encoder (:108-113,159-160); no caller of the
            # reference ever selects it, and it is outside the hot-path scope.
            raise NotImplementedError('condition_mode="add" is not provided by the HIP engine')
        self.image_size = image_size
        self.num_inference_steps = num_inference_steps
        self.condition_mode = condition_mode
        in_channels = 6
        extra = {"allow_unpinned_groupnorm": True} if allow_unpinned_groupnorm else {}  # tiny / base: see unet.py
        self.unet = unet if unet is not None else create_efficient_unet(
            variant=unet_variant, image_size=image_size, in_channels=in_channels, **extra)
        self.scheduler = scheduler if scheduler is not None else LCMScheduler(
            num_train_timesteps=1000, beta_schedule="scaled_linear", prediction_type="epsilon",
            num_inference_steps=num_inference_steps, rescale_betas_zero_snr=True)
        self.compute_dtype = compute_dtype
        object.__setattr__(self, "_t_cache", {})  # (timesteps, batch, device) -> device int64 [steps*B]

    def __getstate__(self):  # copy.deepcopy / pickling: device caches are rebuilt lazily
        st = dict(self.__dict__)
        st["_t_cache"] = {}
        return st

    @property
    def compute_dtype(self) -> Optional[str]:
        return self.unet.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, v: Optional[str]) -> None:
        self.unet.compute_dtype = v

    # ------------------------------------------------------------------ training-side forward (:115-175)
    def forward(self, low_light: torch.Tensor, normal_light: Optional[torch.Tensor] = None,
                timesteps: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                return_dict: bool = True) -> Union[torch.Tensor, Dict[str, torch.Tensor]]:
        """With `normal_light`: q-sample -> denoiser, returning {noise_pred, noise, timesteps}.  When gradients
        are enabled `noise_pred` carries a grad_fn whose backward is the engine's reverse pass
        (llie_unet_backward): `loss.backward()` fills `.grad` of every parameter, so the reference trainer's
        step (trainer.py:269-338: AdamW, GradScaler, clip_grad_norm_, EMA) runs unchanged on top.
        With a `prediction_type="v_prediction"` scheduler the dict also holds the velocity `target`
        (lcm_scheduler.py:282-305; the reference never wires it into `compute_loss`).
        Without `normal_light`: `enhance(low_light)`."""
        if normal_light is None:
            return self.enhance(low_light)
        batch, device = low_light.shape[0], low_light.device
        if timesteps is None:
            timesteps = torch.randint(0, self.scheduler.config.num_train_timesteps, (batch,), device=device)
        if noise is None:
            noise = torch.randn_like(normal_light)
        noisy = self.scheduler.add_noise(normal_light, noise, timesteps)
        noise_pred = self.unet.forward_split(noisy, low_light, timesteps, uniform_t=False)
        if return_dict:
            out = {"noise_pred": noise_pred, "noise": noise, "timesteps": timesteps}
            if getattr(self.scheduler.config, "prediction_type", "epsilon") == "v_prediction":
                out["target"] = self.scheduler.get_velocity(normal_light, noise, timesteps)
            return out
        return noise_pred

    # ------------------------------------------------------------------ inference loop (:177-248)
    @torch.no_grad()
    def enhance(self, low_light: torch.Tensor, num_inference_steps: Optional[int] = None,
                generator: Optional[torch.Generator] = None, return_intermediate: bool = False, *,
                noise: Optional[Union[torch.Tensor, Sequence[torch.Tensor]]] = None,
                return_noise_pred: bool = False) -> Union[torch.Tensor, LowLightDiffusionOutput]:
        """low_light [B,3,S,S] in [-1,1] -> enhanced [B,3,S,S].

        Noise: by default drawn on the device in the reference's order -- the initial latents with
        `generator` (:208-211), then one draw per non-final step from the global generator
        (lcm_scheduler.py:237).  `noise=` (extension) supplies those draws, e.g. CPU-generated ones for
        a bit-comparable run against the CPU reference: a [steps,B,3,S,S] tensor or a list of `steps`
        tensors (entries after the first are the re-noising draws of steps 0..steps-2)."""
        device = low_light.device
        if device.type != "cuda":
            raise RuntimeError("LowLightDiffusion.enhance runs only on a HIP device; there is no CPU fallback")
        b, s = low_light.shape[0], self.image_size
        if tuple(low_light.shape[1:]) != (3, s, s):
            raise ValueError(f"low_light must be [B,3,{s},{s}] (latents are allocated at image_size, "
                             f"low_light_diffusion.py:208-210); got {tuple(low_light.shape)}")
        steps = self.num_inference_steps if num_inference_steps is None else num_inference_steps
        self.scheduler.set_timesteps(steps, device=device)
        ts = self.scheduler._timestep_list
        steps = len(ts)

        if noise is None:
            # drawn straight into the [steps,B,3,S,S] buffer the engine reads (same generator streams as torch.randn)
            noise_t = torch.empty(steps, b, 3, s, s, dtype=torch.float32, device=device)
            noise_t[0].normal_(generator=generator)
            for i in range(1, steps):
                noise_t[i].normal_()
        else:
            noise_t = noise if isinstance(noise, torch.Tensor) else torch.stack([n.to(device) for n in noise])
            noise_t = noise_t.to(device=device, dtype=torch.float32)
            if tuple(noise_t.shape) != (steps, b, 3, s, s):
                raise ValueError(f"noise must be [{steps},{b},3,{s},{s}]")
        noise_t = noise_t.contiguous()

        coefs = (N.StepCoef * steps)(*[self.scheduler.step_coefficients(t) for t in ts])
        tkey = (tuple(ts), b, device.type, device.index)
        t_dev = self._t_cache.get(tkey)  # [steps*B] device timesteps: one H2D copy per (schedule, batch), not per call
        if t_dev is None:
            if len(self._t_cache) > 64:
                self._t_cache.clear()
            t_dev = self._t_cache[tkey] = torch.tensor(ts, dtype=torch.long).repeat_interleave(b).to(device)
        low = low_light.detach().float().contiguous()
        enhanced = torch.empty(b, 3, s, s, dtype=torch.float32, device=device)
        inter = torch.empty(steps, b, 3, s, s, dtype=torch.float32, device=device) if return_intermediate else None
        preds = torch.empty(steps, b, 3, s, s, dtype=torch.float32, device=device) if return_noise_pred else None
        h, ws, nbytes = self.unet._prepare(b, device, enhance_steps=max(steps, 8))
        with torch.cuda.device(device):
            N.check(N.lib().llie_enhance(
                h.h, low.data_ptr(), noise_t.data_ptr(), t_dev.data_ptr(), coefs, steps, enhanced.data_ptr(),
                inter.data_ptr() if inter is not None else None, preds.data_ptr() if preds is not None else None,
                b, ws.data_ptr(), nbytes, torch.cuda.current_stream(device).cuda_stream), "enhance")
        self.scheduler._step_index = steps
        if return_intermediate or return_noise_pred:
            out = LowLightDiffusionOutput(enhanced=enhanced,
                                          intermediate=[inter[i] for i in range(steps)] if inter is not None else None)
            if preds is not None:
                out.noise_pred = [preds[i] for i in range(steps)]
            return out
        return enhanced

    # ------------------------------------------------------------------ loss (:250-277)
    def compute_loss(self, low_light: torch.Tensor, normal_light: torch.Tensor, loss_type: str = "mse", *,
                     use_velocity_target: bool = False) -> torch.Tensor:
        """Regresses the denoiser output against the drawn noise, as the reference does whatever the scheduler's
        prediction type (low_light_diffusion.py:262-275).  `use_velocity_target=True` (extension, needs a
        `prediction_type="v_prediction"` scheduler) regresses against `scheduler.get_velocity` instead -- the v-pred MSE of
        BASELINE config 5, which the reference defines (lcm_scheduler.py:282-305) but never wires into its loss."""
        out = self.forward(low_light, normal_light)
        if use_velocity_target and "target" not in out:
            raise ValueError("use_velocity_target needs a scheduler with prediction_type='v_prediction'")
        pred, noise = out["noise_pred"], (out["target"] if use_velocity_target else out["noise"])
        if loss_type == "mse":
            return F.mse_loss(pred, noise)
        if loss_type == "huber":
            return F.huber_loss(pred, noise)
        if loss_type == "l1":
            return F.l1_loss(pred, noise)
        raise ValueError(f"Unknown loss type: {loss_type}")

    def get_model_size(self) -> Dict[str, float]:
        return self.unet.get_memory_footprint()


def normalize_image(x: torch.Tensor) -> torch.Tensor:
    """[0,1] -> [-1,1] (low_light_diffusion.py:412-414)."""
    return x * 2 - 1


def denormalize_image(x: torch.Tensor) -> torch.Tensor:
    """[-1,1] -> [0,1] (low_light_diffusion.py:417-419)."""
    return (x + 1) / 2
