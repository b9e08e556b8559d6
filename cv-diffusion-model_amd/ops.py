"""`torch.ops.llie.*`: the engine's entry points registered as PyTorch custom operators (SURVEY.md 8b asks for the
C ABI to be reachable "via PyTorch-ROCm custom ops").  The operators are thin: they look a model up in a registry and
call the same ctypes path as the module methods, so eager code, `torch.library.opcheck`-style tooling and exporters
see named ops with fake (meta) implementations instead of opaque Python.

    mid = register_model(model)                       # model: LowLightDiffusion (or anything with .unet / .enhance);
                                                      # register_model(model, key=7) pins the slot for exported graphs
    y   = torch.ops.llie.enhance(mid, low_light, noise, 4)          # [B,3,S,S] -> [B,3,S,S]
    eps = torch.ops.llie.unet_forward(mid, latents, cond, t)        # noise prediction, per-sample timesteps
    x   = torch.ops.llie.lcm_step(model_output, sample, noise, sqrt_a_t, sqrt_b_t, sqrt_a_prev, sqrt_b_prev, last, vpred)

No CPU implementation is registered: on CPU tensors the ops raise, like the module methods do.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional

import torch

from . import _native as N

_MODELS: Dict[int, "weakref.ReferenceType"] = {}


_NEXT_KEY = [1]


def register_model(model, key: Optional[int] = None) -> int:
    """Make `model` addressable from the custom ops and return the integer the ops take as `model_id` (the registry holds a
    weak reference).  `key` chooses that integer: a graph exported with `model_id = 7` baked in as a constant runs in another
    process after `register_model(its_model, key=7)` -- the id is a slot of the caller's choosing, not the address of a
    Python object.  Without `key` the next free small integer is used; registering the same model again returns its slot."""
    if key is None:
        for k, ref in _MODELS.items():
            if ref() is model:
                return k
        while _NEXT_KEY[0] in _MODELS and _MODELS[_NEXT_KEY[0]]() is not None:
            _NEXT_KEY[0] += 1
        key = _NEXT_KEY[0]
        _NEXT_KEY[0] += 1
    key = int(key)
    cur = _MODELS.get(key)
    if cur is not None and cur() is not None and cur() is not model:
        raise ValueError(f"llie: slot {key} already holds another live model")
    _MODELS[key] = weakref.ref(model)
    return key


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
