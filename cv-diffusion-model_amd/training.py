"""The optimiser side of the reference trainer's inner loop (src/training/trainer.py:281-324) on the HIP engine.

`FusedAdamW`  -- `torch.optim.AdamW` (trainer.py:163-168) whose `step()` also does what the trainer does around it:
                 `scaler.unscale_` / the 1 / world-size of the gradient average (`grad_scale`), `clip_grad_norm_(params,
                 gradient_clip)` (`max_grad_norm`, trainer.py:296-299,310-313) and `EMAModel.update` (`ema_decay`,
                 trainer.py:98-104) -- in three launches over all parameter tensors (llie_optimizer_step, csrc/optim.hip)
                 instead of ~40 launches and several milliseconds of host time on 321 tensors.  It is a
                 `torch.optim.Optimizer`: LR schedulers (`CosineAnnealingLR`, `OneCycleLR`, trainer.py:162-175) drive
                 `param_groups[0]["lr"]`, `state_dict()` has torch.optim.AdamW's layout (the trainer's checkpoint,
                 trainer.py:418-434).
`TrainStep`   -- `compute_loss -> backward -> gradient all-reduce -> clip -> AdamW -> EMA` without the autograd graph: the
                 engine's backward pass writes one flat gradient buffer and the optimiser reads it in place (no per-parameter
                 `.grad` views, no AccumulateGrad nodes).  Same numbers as the autograd path (tests/test_gpu_round4.py).

There is no CPU fallback: both need the HIP library and parameters on a HIP device.
"""
import ctypes as C
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import _native as N


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, *, max_grad_norm: Optional[float] = None, ema_decay: Optional[float] = None,
                 skip_nonfinite: bool = False):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("FusedAdamW: lr, eps, weight_decay >= 0 and 0 <= beta < 1 (torch.optim.AdamW's checks)")
        if ema_decay is not None and not (0 <= ema_decay <= 1):
            raise ValueError("ema_decay must be in [0, 1]")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamW takes one parameter group (the reference trainer's model.parameters(), trainer.py:164)")
        ps = self.param_groups[0]["params"]
        if not ps:
            raise ValueError("no parameters")
        dev = ps[0].device
        for p in ps:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("FusedAdamW: fp32 contiguous parameters on one device (the engine's masters)")
        if dev.type != "cuda":
            raise RuntimeError("FusedAdamW runs only on a HIP device; there is no CPU fallback")
        self.max_grad_norm = max_grad_norm
        self.ema_decay = ema_decay
        self.skip_nonfinite = skip_nonfinite
        self._dev = dev
        self._numel = [p.numel() for p in ps]
        total = sum(self._numel)
        self._m = torch.zeros(total, dtype=torch.float32, device=dev)
        self._v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._ema = torch.cat([p.detach().reshape(-1) for p in ps]) if ema_decay is not None else None  # EMAModel.__init__: a clone
        self._stats = torch.zeros(3, dtype=torch.float32, device=dev)
        self._step = 0
        self._native = None   # llie_optimizer*
        self._layout = None   # (param pointers, gradient offsets) the native tables were built for
        o = 0
        for p, n in zip(ps, self._numel):
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self._m[o:o + n].view_as(p), "exp_avg_sq": self._v[o:o + n].view_as(p)}
            o += n

    # ------------------------------------------------------------------ native tables
    def _bind(self, offsets: List[int]) -> None:
        ps = self.param_groups[0]["params"]
        layout = (tuple(p.data_ptr() for p in ps), tuple(offsets))
        if self._native is not None and layout == self._layout:
            return
        self._release()
        arr = (N.OptTensor * len(ps))()
        o = 0
        for i, (p, n) in enumerate(zip(ps, self._numel)):
            e = self._ema.data_ptr() + 4 * o if self._ema is not None else None
            arr[i] = N.OptTensor(p.data_ptr(), self._m.data_ptr() + 4 * o, self._v.data_ptr() + 4 * o, e, offsets[i], n)
            o += n
        out = C.c_void_p()
        with torch.cuda.device(self._dev):
            N.check(N.lib().llie_optimizer_create(arr, len(ps), C.byref(out)), "FusedAdamW")
        self._native, self._layout = out, layout

    def _release(self) -> None:
        if getattr(self, "_native", None) is not None:
            torch.cuda.synchronize(self._dev)  # a step reading the tables may still be in flight
            N.lib().llie_optimizer_destroy(self._native)
            self._native = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ the step
    def _launch(self, grad_base: int, grad_scale: float) -> None:
        g = self.param_groups[0]
        self._step += 1
        h = N.OptHyper(float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                       float(self.max_grad_norm) if self.max_grad_norm else 0.0,
                       float(self.ema_decay) if self.ema_decay is not None else -1.0, float(grad_scale), self._step,
                       1 if self.skip_nonfinite else 0)
        with torch.cuda.device(self._dev):
            N.check(N.lib().llie_optimizer_step(self._native, grad_base, C.byref(h), self._stats.data_ptr(),
                                                torch.cuda.current_stream(self._dev).cuda_stream), "FusedAdamW.step")

    @torch.no_grad()
    def step_flat(self, flat: torch.Tensor, offsets: List[int], grad_scale: float = 1.0) -> torch.Tensor:
        """One update from a flat fp32 gradient buffer (parameter i at `flat[offsets[i]:]`, e.g. what llie_unet_backward
        writes).  Returns the gradient norm before clipping as a device scalar (what clip_grad_norm_ returns)."""
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.device != self._dev:
            raise ValueError("flat gradients: contiguous fp32 on the parameters' device")
        if len(offsets) != len(self._numel) or any(o < 0 or o + n > flat.numel() for o, n in zip(offsets, self._numel)):
            raise ValueError("gradient offsets do not fit the flat buffer")
        self._bind(list(offsets))
        self._launch(flat.data_ptr(), grad_scale)
        return self._stats[0]

    @torch.no_grad()
    def step(self, closure=None, *, grad_scale: float = 1.0):
        """`clip_grad_norm_` (if max_grad_norm) + `AdamW.step` + EMA update (if ema_decay) on the `.grad` of every parameter.
        Every parameter must have a gradient (the engine's backward pass always writes all of them)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps = self.param_groups[0]["params"]
        grads = []
        for p in ps:
            g = p.grad
            if g is None:
                raise RuntimeError("FusedAdamW.step: a parameter has no gradient")
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != self._dev:
                g = g.to(device=self._dev, dtype=torch.float32).contiguous()
            grads.append(g)
        ptrs = [g.data_ptr() for g in grads]
        base = min(ptrs)
        # offsets relative to the lowest gradient: with the engine's backward pass these are the views of one flat buffer
        # and never change; gradients from elsewhere just rebuild the tables when their relative placement moves
        self._bind([(q - base) // 4 for q in ptrs])
        self._launch(base, grad_scale)
        self._keep = grads  # alive until the next step's launch is queued behind this one
        return loss

    def grad_norm(self) -> torch.Tensor:
        """Norm of the (scaled) gradients of the last step, before clipping: device scalar."""
        return self._stats[0]

    def last_step_skipped(self) -> bool:
        return bool(self._stats[2].item())

    # ------------------------------------------------------------------ EMA (trainer.py:86-118)
    def ema_tensors(self) -> List[torch.Tensor]:
        """Shadow weights as views shaped like the parameters, in parameter order."""
        if self._ema is None:
            raise RuntimeError("constructed without ema_decay")
        out, o = [], 0
        for p, n in zip(self.param_groups[0]["params"], self._numel):
            out.append(self._ema[o:o + n].view_as(p))
            o += n
        return out

    # ------------------------------------------------------------------ checkpoint layout of torch.optim.AdamW
    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        sd = super().state_dict()
        if self._ema is not None:
            sd["ema_shadow_flat"] = self._ema.clone()
        return sd

    def load_state_dict(self, state_dict) -> None:
        sd = dict(state_dict)
        ema = sd.pop("ema_shadow_flat", None)
        super().load_state_dict(sd)
        ps = self.param_groups[0]["params"]
        o, step = 0, 0
        for p, n in zip(ps, self._numel):
            st = self.state.get(p, {})
            if "exp_avg" in st:
                self._m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self._v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, int(float(st.get("step", 0))))
            self.state[p] = {"step": torch.tensor(float(step)), "exp_avg": self._m[o:o + n].view_as(p), "exp_avg_sq": self._v[o:o + n].view_as(p)}
            o += n
        self._step = step
        if ema is not None and self._ema is not None:
            self._ema.copy_(ema.to(self._dev))


class TrainStep:
    """One optimisation step of the reference trainer (trainer.py:281-324) on the engine, without autograd:
    q-sample (low_light_diffusion.py:140-160) -> llie_unet_train_forward -> loss and d(loss)/d(eps) -> llie_unet_backward into a
    persistent flat buffer -> one all-reduce of that buffer over the ranks -> FusedAdamW.step_flat (clip, AdamW, EMA).
    `loss_type` / `use_velocity_target` as LowLightDiffusion.compute_loss.  Returns the loss (device scalar)."""

    def __init__(self, model, optimizer: FusedAdamW, loss_type: str = "mse", use_velocity_target: bool = False, group=None):
        if loss_type not in ("mse", "huber", "l1"):
            raise ValueError(f"Unknown loss type: {loss_type}")
        self.model, self.opt, self.loss_type, self.velocity, self.group = model, optimizer, loss_type, use_velocity_target, group
        unet = model.unet
        index = {id(p): i for i, (_, p) in enumerate(unet._ordered_params())}  # engine (llie_param_info) order
        have = optimizer.param_groups[0]["params"]
        if len(index) != len(have) or any(id(p) not in index for p in have):
            raise ValueError("TrainStep: the optimiser must hold exactly model.unet's parameters (FusedAdamW(model.parameters(), ...))")
        self._order = [index[id(p)] for p in have]
        if use_velocity_target and getattr(model.scheduler.config, "prediction_type", "epsilon") != "v_prediction":
            raise ValueError("use_velocity_target needs a scheduler with prediction_type='v_prediction'")
        self._flat = None
        self._ws = None

    @torch.no_grad()
    def __call__(self, low_light: torch.Tensor, normal_light: torch.Tensor, timesteps: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        from .unet import resolve_compute_dtype
        model, unet = self.model, self.model.unet
        b, dev = low_light.shape[0], low_light.device
        s = unet.config.image_size
        if tuple(low_light.shape[1:]) != (3, s, s) or tuple(normal_light.shape) != tuple(low_light.shape):
            raise ValueError(f"low_light / normal_light must be [B,3,{s},{s}]")
        if timesteps is None:
            timesteps = torch.randint(0, model.scheduler.config.num_train_timesteps, (b,), device=dev)
        if noise is None:
            noise = torch.randn_like(normal_light)
        noisy = model.scheduler.add_noise(normal_light, noise, timesteps).float().contiguous()
        target = model.scheduler.get_velocity(normal_light, noise, timesteps) if self.velocity else noise
        cond = low_light.detach().float().contiguous()
        t = timesteps.to(device=dev, dtype=torch.long).contiguous()
        h = unet._handle(resolve_compute_dtype(unet.compute_dtype))
        nbytes = h.train_workspace_bytes(b)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if self._flat is None or self._flat.numel() != h.grad_numel() or self._flat.device != dev:
            self._flat = torch.empty(h.grad_numel(), dtype=torch.float32, device=dev)
            offs = h.grad_offsets()
            self._offsets = [offs[i] for i in self._order]
        eps = torch.empty(b, unet.config.out_channels, s, s, dtype=torch.float32, device=dev)
        L = N.lib()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            N.check(L.llie_unet_train_forward(h.h, noisy.data_ptr(), cond.data_ptr(), t.data_ptr(), eps.data_ptr(), b,
                                              self._ws.data_ptr(), nbytes, st), "EfficientUNet.forward (training)")
            diff = eps - target
            n = diff.numel()
            if self.loss_type == "mse":          # F.mse_loss: mean(d^2); d/d eps = 2 d / n
                loss = (diff * diff).mean()
                d_eps = diff * (2.0 / n)
            elif self.loss_type == "l1":         # F.l1_loss: mean|d|; sign(d) / n
                loss = diff.abs().mean()
                d_eps = torch.sign(diff) / n
            else:                                # F.huber_loss(delta=1): 0.5 d^2 inside, |d| - 0.5 outside; clamp(d, -1, 1) / n
                a = diff.abs()
                loss = torch.where(a < 1.0, 0.5 * diff * diff, a - 0.5).mean()
                d_eps = diff.clamp(-1.0, 1.0) / n
            N.check(L.llie_unet_backward(h.h, d_eps.data_ptr(), self._flat.data_ptr(), b, self._ws.data_ptr(), nbytes, st),
                    "EfficientUNet.backward")
        scale = 1.0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self._flat, group=self.group)  # one collective over all gradients; the average rides on grad_scale
            scale = 1.0 / dist.get_world_size(self.group)
        self.opt.step_flat(self._flat, self._offsets, grad_scale=scale)
        return loss
