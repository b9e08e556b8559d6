"""LCMScheduler with the reference's interface (src/models/lcm_scheduler.py:34-305), no `diffusers`.

Host logic (beta / alpha-bar tables, timestep selection, per-step scalar coefficients) is plain
PyTorch-CPU arithmetic in fp32, issued in the reference's order so the tables are bit-identical.
Tensor work (`step`, `add_noise`, `get_velocity`) runs in libllie_hip.so on the tensors' HIP device;
there is no CPU path for it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import torch

from . import _native as N


@dataclass
class LCMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: tensors must live on a HIP device (got '{t.device}'); there is no CPU fallback")


class LCMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "epsilon",
                 timestep_spacing: str = "leading", rescale_betas_zero_snr: bool = False,
                 num_inference_steps: int = 4, original_inference_steps: int = 50, lcm_origin_steps: int = 50):
        # what diffusers' @register_to_config exposes as self.config (lcm_scheduler.py:53-66)
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
            rescale_betas_zero_snr=rescale_betas_zero_snr, num_inference_steps=num_inference_steps,
            original_inference_steps=original_inference_steps, lcm_origin_steps=lcm_origin_steps)
        if beta_schedule == "linear":  # :77-87
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            self.betas = self._cosine_betas(num_train_timesteps)
        else:
            raise ValueError(f"Unknown beta schedule: {beta_schedule}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        if rescale_betas_zero_snr:  # :116-129
            root = self.alphas_cumprod.sqrt()
            first, last = root[0].clone(), root[-1].clone()
            root = (root - last) * (first / (first - last))
            self.alphas_cumprod = root ** 2
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None
        self._step_index = None
        self._timestep_list: List[int] = []
        self._acp_dev = {}
        self._ts_dev = {}

    @staticmethod
    def _cosine_betas(n: int, s: float = 0.008) -> torch.Tensor:  # :107-114
        x = torch.linspace(0, n, n + 1)
        acp = torch.cos(((x / n) + s) / (1 + s) * math.pi * 0.5) ** 2
        acp = acp / acp[0]
        return torch.clip(1 - (acp[1:] / acp[:-1]), 0, 0.999)

    # ------------------------------------------------------------------ timestep selection (:131-167)
    def set_timesteps(self, num_inference_steps: int = 4, device: Union[str, torch.device] = "cpu",
                      original_inference_steps: Optional[int] = None) -> None:
        self.num_inference_steps = num_inference_steps
        if original_inference_steps is None:
            original_inference_steps = self.config.original_inference_steps
        c = self.config.num_train_timesteps // original_inference_steps
        origin = torch.arange(1, original_inference_steps + 1) * c - 1
        skipping_step = len(origin) // num_inference_steps
        ts = origin[::skipping_step][:num_inference_steps].flip(0)  # slice step 0 -> ValueError, like the reference
        self._timestep_list = [int(v) for v in ts.tolist()]
        # device copies are cached per (schedule, device): a pageable host-to-device copy blocks the host until the stream
        # has drained, i.e. every `enhance` call would wait for the previous one and leave the GPU idle while Python
        # prepares the next launch (measured: 0.7 ms of exposed host time per call, 10 % of a B=1 call)
        dev = torch.device(device)
        index = dev.index
        if dev.type == "cuda" and index is None:  # "cuda" means the *current* device, which set_device can change
            index = torch.cuda.current_device()
        key = (tuple(self._timestep_list), dev.type, index)
        cached = self._ts_dev.get(key)
        if cached is None:
            if len(self._ts_dev) > 64:
                self._ts_dev.clear()
            cached = self._ts_dev[key] = ts.to(dev)
        # a fresh tensor per call like the reference (:163): a caller editing `timesteps` in place must not poison the
        # cache.  Device-to-device copy, asynchronous, no host sync.
        self.timesteps = cached.clone()
        self._step_index = 0
        if self.sigmas.device != dev:
            self.sigmas = self.sigmas.to(dev)

    def _get_prev_timestep(self, timestep: int) -> int:  # :169-174, without the device round trip
        idx = self._timestep_list.index(int(timestep))
        return self._timestep_list[idx + 1] if idx + 1 < len(self._timestep_list) else 0

    def step_coefficients(self, timestep: int) -> N.StepCoef:
        """Scalars of one step, computed with the reference's 0-d fp32 tensor arithmetic (:208-212)."""
        t = int(timestep)
        prev_t = self._get_prev_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t > 0 else self.final_alpha_cumprod
        ptype = self.config.prediction_type
        if ptype not in ("epsilon", "v_prediction"):
            raise ValueError(f"Unknown prediction type: {ptype}")
        return N.StepCoef(float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5),
                          int(prev_t == 0), int(ptype == "v_prediction"))

    # ------------------------------------------------------------------ tensor ops on the device
    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor,
             generator: Optional[torch.Generator] = None, return_dict: bool = True,
             noise: Optional[torch.Tensor] = None) -> Union[LCMSchedulerOutput, Tuple]:
        """lcm_scheduler.py:176-253.  Like the reference, `generator` is accepted and ignored: the
        re-noising draw comes from the global generator of the sample's device (:237).  `noise=` is an
        extension that lets a caller supply that draw."""
        _require_cuda(sample, "LCMScheduler.step")
        if self._step_index is None:
            self._step_index = 0
        coef = self.step_coefficients(int(timestep))
        sample_c = sample.detach().float().contiguous()
        mo = model_output.detach().float().contiguous()
        if not coef.is_last and noise is None:
            noise = torch.randn_like(sample_c)
        prev = torch.empty_like(sample_c)
        x0 = torch.empty_like(sample_c)
        with torch.cuda.device(sample.device):
            N.check(N.lib().llie_lcm_step(mo.data_ptr(), sample_c.data_ptr(),
                                          None if coef.is_last else noise.float().contiguous().data_ptr(),
                                          prev.data_ptr(), x0.data_ptr(), None, sample_c.numel(), coef,
                                          torch.cuda.current_stream(sample.device).cuda_stream), "LCMScheduler.step")
        self._step_index += 1
        if return_dict:
            return LCMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)
        return (prev, x0)

    def _acp_on(self, device: torch.device) -> torch.Tensor:
        key = (device.type, device.index)
        if key not in self._acp_dev:
            self._acp_dev[key] = self.alphas_cumprod.to(device=device, dtype=torch.float32).contiguous()
        return self._acp_dev[key]

    def _noise_op(self, a: torch.Tensor, b: torch.Tensor, timesteps: torch.Tensor, velocity: int) -> torch.Tensor:
        _require_cuda(a, "LCMScheduler.add_noise/get_velocity")
        a_c, b_c = a.detach().float().contiguous(), b.detach().float().contiguous()
        n_table = int(self.alphas_cumprod.numel())
        if timesteps.device.type == "cpu" and timesteps.numel() and (int(timesteps.min()) < -n_table or int(timesteps.max()) >= n_table):
            # the reference indexes a tensor with the timesteps (lcm_scheduler.py:268) and raises; device-resident
            # timesteps cannot be checked without a sync: the kernel turns an out-of-range one into NaN output instead
            raise IndexError(f"timestep out of range for a table of {n_table} entries")
        t = timesteps.to(device=a.device, dtype=torch.long).contiguous()
        if timesteps.device.type == "cpu" and timesteps.numel() and int(timesteps.min()) < 0:
            t = torch.where(t < 0, t + n_table, t)  # negative indices wrap like tensor indexing does
        out = torch.empty_like(a_c)
        batch = a_c.shape[0]
        with torch.cuda.device(a.device):
            N.check(N.lib().llie_add_noise(a_c.data_ptr(), b_c.data_ptr(), t.data_ptr(), self._acp_on(a.device).data_ptr(), n_table,
                                           out.data_ptr(), batch, a_c.numel() // batch, velocity,
                                           torch.cuda.current_stream(a.device).cuda_stream), "add_noise")
        return out

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """x_t = sqrt(abar_t) x_0 + sqrt(1-abar_t) noise (:255-280)."""
        return self._noise_op(original_samples, noise, timesteps, 0)

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """v = sqrt(abar_t) noise - sqrt(1-abar_t) sample (:282-305)."""
        return self._noise_op(sample, noise, timesteps, 1)


def get_lcm_timesteps(num_inference_steps: int = 4, num_train_timesteps: int = 1000,
                      original_inference_steps: int = 50) -> List[int]:
    """List form of the selection rule (lcm_scheduler.py:421-442)."""
    c = num_train_timesteps // original_inference_steps
    origin = [(i + 1) * c - 1 for i in range(original_inference_steps)]
    skip = len(origin) // num_inference_steps
    return list(reversed(origin[::skip][:num_inference_steps]))


class LCMDenoisingLoop:
    """The deployment loop's scheduler semantics (src/export/android_pipeline.py:191-277) as a selectable
    mode: scaled-linear alpha-bar table in float64 WITHOUT the zero-SNR rescale, and x0 clamped to [-1, 1]
    before re-noising.  Interface as the reference class (`timesteps`, `alphas_cumprod`, `add_noise`,
    `step`), with device tensors instead of numpy arrays; it also answers the calls
    `LowLightDiffusion.enhance` makes on its scheduler, so `LowLightDiffusion(scheduler=LCMDenoisingLoop())`
    runs the whole loop with these semantics.

    The per-step scalars are formed in float64 like the reference and rounded once to fp32 for the
    kernel; the reference's own result dtype depends on numpy's promotion rules (float64 under
    NumPy >= 2), so parity is to fp32 rounding, not bitwise."""

    def __init__(self, num_train_timesteps: int = 1000, num_inference_steps: int = 4,
                 beta_start: float = 0.00085, beta_end: float = 0.012):
        import numpy as np
        self.num_train_timesteps = num_train_timesteps
        self.num_inference_steps = num_inference_steps
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps) ** 2
        self.alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(self.alphas)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type="epsilon",
                                      original_inference_steps=50)
        self._acp_dev = {}
        self._step_index = None
        self.set_timesteps(num_inference_steps)

    def _get_lcm_timesteps(self):
        import numpy as np
        c = self.num_train_timesteps // 50
        lcm = np.arange(1, 51) * c - 1
        skip = len(lcm) // self.num_inference_steps
        return lcm[::skip][:self.num_inference_steps][::-1].copy()

    def set_timesteps(self, num_inference_steps: int = 4, device=None) -> None:
        self.num_inference_steps = num_inference_steps
        self.timesteps = self._get_lcm_timesteps()
        self._timestep_list = [int(v) for v in self.timesteps]
        self._step_index = 0

    def step_coefficients(self, timestep: int) -> N.StepCoef:
        t = int(timestep)
        idx = self._timestep_list.index(t)
        prev_t = self._timestep_list[idx + 1] if idx + 1 < len(self._timestep_list) else 0
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev_t] if prev_t > 0 else self.alphas_cumprod[0])
        return N.StepCoef(math.sqrt(a_t), math.sqrt(1 - a_t), math.sqrt(a_p), math.sqrt(1 - a_p), int(prev_t == 0), 0, 1)

    def step(self, noise_pred: torch.Tensor, timestep: int, sample: torch.Tensor,
             noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """android_pipeline.py:240-265 -> the next sample (x0 after the last timestep).  `noise=` supplies
        the re-noising draw (the reference draws it with np.random.randn, :262)."""
        _require_cuda(sample, "LCMDenoisingLoop.step")
        coef = self.step_coefficients(int(timestep))
        sample_c = sample.detach().float().contiguous()
        mo = noise_pred.detach().float().contiguous()
        if not coef.is_last and noise is None:
            noise = torch.randn_like(sample_c)
        prev = torch.empty_like(sample_c)
        with torch.cuda.device(sample.device):
            N.check(N.lib().llie_lcm_step(mo.data_ptr(), sample_c.data_ptr(),
                                          None if coef.is_last else noise.float().contiguous().data_ptr(),
                                          prev.data_ptr(), None, None, sample_c.numel(), coef,
                                          torch.cuda.current_stream(sample.device).cuda_stream), "LCMDenoisingLoop.step")
        return prev

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timestep: int) -> torch.Tensor:
        """android_pipeline.py:228-238 (one scalar timestep for the whole batch)."""
        _require_cuda(original, "LCMDenoisingLoop.add_noise")
        a, b = original.detach().float().contiguous(), noise.detach().float().contiguous()
        key = (a.device.type, a.device.index)
        if key not in self._acp_dev:
            self._acp_dev[key] = torch.from_numpy(self.alphas_cumprod).to(device=a.device, dtype=torch.float32).contiguous()
        n_table = len(self.alphas_cumprod)
        if not -n_table <= int(timestep) < n_table:
            raise IndexError(f"timestep {int(timestep)} out of range for a table of {n_table} entries")  # numpy indexing in the reference
        t = torch.full((a.shape[0],), int(timestep) % n_table, dtype=torch.long, device=a.device)
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            N.check(N.lib().llie_add_noise(a.data_ptr(), b.data_ptr(), t.data_ptr(), self._acp_dev[key].data_ptr(), n_table,
                                           out.data_ptr(), a.shape[0], a[0].numel(), 0,
                                           torch.cuda.current_stream(a.device).cuda_stream), "LCMDenoisingLoop.add_noise")
        return out
