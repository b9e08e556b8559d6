"""`torch.ops.llie.*`: the engine's entry points registered as PyTorch custom operators (SURVEY.md 8b asks for the
C ABI to be reachable "via PyTorch-ROCm custom ops").  The operators are thin: they look a model up in a registry and
call the same ctypes path as the module methods, so eager code, `torch.library.opcheck`-style tooling and exporters
see named ops with fake (meta) implementations instead of opaque Python.

    mid = register_model(model)                       # model: LowLightDiffusion (or anything with .unet / .enhance)
    y   = torch.ops.llie.enhance(mid, low_light, noise, 4)          # [B,3,S,S] -> [B,3,S,S]
    eps = torch.ops.llie.unet_forward(mid, latents, cond, t)        # noise prediction, per-sample timesteps
    x   = torch.ops.llie.lcm_step(model_output, sample, noise, sqrt_a_t, sqrt_b_t, sqrt_a_prev, sqrt_b_prev, last, vpred)

No CPU implementation is registered: on CPU tensors the ops raise, like the module methods do.
"""
from __future__ import annotations

import weakref
from typing import Dict

import torch

from . import _native as N

_MODELS: Dict[int, "weakref.ReferenceType"] = {}


def register_model(model) -> int:
    """Make `model` addressable from the custom ops; returns its id (the registry holds a weak reference)."""
    mid = id(model)
    _MODELS[mid] = weakref.ref(model)
    return mid


def _model(mid: int):
    ref = _MODELS.get(int(mid))
    m = ref() if ref is not None else None
    if m is None:
        raise RuntimeError(f"llie: no live model registered under id {mid}; call register_model(model) first")
    return m


@torch.library.custom_op("llie::enhance", mutates_args=(), device_types="cuda")
def enhance(model_id: int, low_light: torch.Tensor, noise: torch.Tensor, num_inference_steps: int) -> torch.Tensor:
    return _model(model_id).enhance(low_light, num_inference_steps, noise=noise)


@enhance.register_fake
def _(model_id, low_light, noise, num_inference_steps):
    return low_light.new_empty(low_light.shape, dtype=torch.float32)


@torch.library.custom_op("llie::unet_forward", mutates_args=(), device_types="cuda")
def unet_forward(model_id: int, latents: torch.Tensor, cond: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    with torch.no_grad():
        return _model(model_id).unet.forward_split(latents, cond, timesteps)


@unet_forward.register_fake
def _(model_id, latents, cond, timesteps):
    return latents.new_empty(latents.shape, dtype=torch.float32)


@torch.library.custom_op("llie::lcm_step", mutates_args=(), device_types="cuda")
def lcm_step(model_output: torch.Tensor, sample: torch.Tensor, noise: torch.Tensor, sqrt_alpha_t: float, sqrt_beta_t: float,
             sqrt_alpha_prev: float, sqrt_beta_prev: float, is_last: bool, v_prediction: bool) -> torch.Tensor:
    coef = N.StepCoef(sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, sqrt_beta_prev, int(is_last), int(v_prediction), 0)
    mo, x = model_output.float().contiguous(), sample.float().contiguous()
    prev = torch.empty_like(x)
    with torch.cuda.device(x.device):
        N.check(N.lib().llie_lcm_step(mo.data_ptr(), x.data_ptr(), None if is_last else noise.float().contiguous().data_ptr(),
                                      prev.data_ptr(), None, None, x.numel(), coef,
                                      torch.cuda.current_stream(x.device).cuda_stream), "llie::lcm_step")
    return prev


@lcm_step.register_fake
def _(model_output, sample, noise, sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, sqrt_beta_prev, is_last, v_prediction):
    return sample.new_empty(sample.shape, dtype=torch.float32)
