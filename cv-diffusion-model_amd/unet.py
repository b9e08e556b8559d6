"""Host-side mirror of the reference's denoiser interface, backed by libllie_hip.so.

Same names, constructor arguments, `state_dict` keys/shapes and error behaviour as
`src/models/efficient_unet.py` of the reference (EfficientUNetConfig :24-57, EfficientUNet :387-628,
create_efficient_unet :631-692, InvertedResidualBlock :134-236, LinearAttention :239-308,
Downsample :360-372, Upsample :375-384) -- but no layer here computes anything in PyTorch: the
modules only *hold* the parameters (so `.state_dict()`, `.parameters()`, `.to()`, EMA and optimisers
work as usual) and `forward` hands raw device pointers to the C ABI.  There is no CPU fallback.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _native as N


@dataclass
class EfficientUNetConfig:
    """Field-for-field the reference's dataclass (efficient_unet.py:24-57)."""
    in_channels: int = 3
    out_channels: int = 3
    base_channels: int = 32
    channel_multipliers: Tuple[int, ...] = (1, 2, 4, 8)
    attention_resolutions: Tuple[int, ...] = (16, 8)
    num_attention_heads: int = 4
    use_linear_attention: bool = True
    num_res_blocks: int = 2
    expansion_ratio: int = 4
    use_se: bool = True
    se_ratio: float = 0.25
    time_embed_dim: int = 128
    dropout: float = 0.0
    quantization_friendly: bool = True
    image_size: int = 256
    # extension: build topologies whose GroupNorm(min(32, C), C) the reference cannot construct (tiny, base) with
    # groups = largest divisor of C <= 32.  Nothing can pin their outputs (parity-unpinned); inference only.
    allow_unpinned_groupnorm: bool = False


class _Node(nn.Module):
    """Plain container used to reproduce the reference's nested module names."""


def _register(root: nn.Module, key: str, param: nn.Parameter) -> None:
    mod = root
    parts = key.split(".")
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


def _default_init_(params: Dict[str, nn.Parameter]) -> None:
    """PyTorch's default initialisers for the layers the reference instantiates: Conv2d/Linear
    kaiming_uniform(a=sqrt(5)) weights and U(+-1/sqrt(fan_in)) biases, GroupNorm ones/zeros."""
    with torch.no_grad():
        for key, p in params.items():
            stem, leaf = key.rsplit(".", 1) if "." in key else ("", key)
            if leaf == "weight":
                if p.dim() == 1:
                    p.fill_(1.0)
                else:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            else:  # bias
                w = params.get((stem + "." if stem else "") + "weight")
                if w is None or w.dim() == 1:
                    p.zero_()
                else:
                    fan_in = w[0].numel()
                    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
                    p.uniform_(-bound, bound)


def resolve_compute_dtype(explicit: Optional[str]) -> int:
    """fp32 unless asked otherwise; `torch.autocast("cuda", dtype=...)` -- the reference's only working
    reduced-precision path (trainer.py:285-291; model.half() is broken there) -- selects
    the matching fp16/bf16 engine."""
    if explicit is not None:
        return N.dtype_code(explicit)
    if torch.is_autocast_enabled("cuda"):
        return N.dtype_code(torch.get_autocast_dtype("cuda"))
    return N.LLIE_F32


def _grad_views(flat: torch.Tensor, handle: "N.Handle", param_list) -> Tuple[torch.Tensor, ...]:
    """Per-parameter views (state_dict shapes) of the engine's flat fp32 gradient buffer."""
    offs = handle.grad_offsets()
    out = []
    for (key, shape), o in zip(param_list, offs):
        n = 1
        for d in shape:
            n *= d
        out.append(flat[o:o + n].view(shape))
    return tuple(out)


class _ModuleFn(torch.autograd.Function):
    """Single operator with autograd: forward = llie_module_forward, backward = llie_module_backward (which
    re-runs the forward with its activations kept, then the reverse pass)."""

    @staticmethod
    def forward(ctx, mod, out_shape, x, temb, *params):
        ctx.mod = mod
        ctx.has_temb = temb is not None
        ctx.save_for_backward(x, temb if temb is not None else x.new_empty(0))
        return mod._run_module_raw(x, temb, out_shape)

    @staticmethod
    def backward(ctx, dy):
        x, temb = ctx.saved_tensors
        dx, dtemb, grads = ctx.mod._run_module_backward(x, temb if ctx.has_temb else None, dy)
        return (None, None, dx, dtemb) + grads


class _UNetFn(torch.autograd.Function):
    """EfficientUNet.forward with autograd: the forward keeps its activations in a workspace owned by this
    node, the backward is llie_unet_backward (parameter gradients only -- the network input is data)."""

    @staticmethod
    def forward(ctx, unet, latents, cond, t, *params):
        h = unet._handle(resolve_compute_dtype(unet.compute_dtype))
        dev = latents.device
        b, s = latents.shape[0], unet.config.image_size
        nbytes = h.train_workspace_bytes(b)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(b, unet.config.out_channels, s, s, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(N.lib().llie_unet_train_forward(h.h, latents.data_ptr(), cond.data_ptr(), t.data_ptr(), out.data_ptr(), b,
                                                    ws.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream),
                    "EfficientUNet.forward (training)")
        ctx.unet, ctx.h, ctx.ws, ctx.nbytes, ctx.batch = unet, h, ws, nbytes, b
        ctx.keep = (latents, cond, t)  # the backward pass reads them again (input conv, time embedding)
        return out

    @staticmethod
    def backward(ctx, d_eps):
        unet, h, dev = ctx.unet, ctx.h, d_eps.device
        d_eps = d_eps.detach().float().contiguous()
        flat = torch.empty(h.grad_numel(), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(N.lib().llie_unet_backward(h.h, d_eps.data_ptr(), flat.data_ptr(), ctx.batch, ctx.ws.data_ptr(), ctx.nbytes,
                                               torch.cuda.current_stream(dev).cuda_stream), "EfficientUNet.backward")
        ctx.ws = None
        return (None, None, None, None) + _grad_views(flat, h, unet._param_list)


class _NativeModule(nn.Module):
    """Parameters live in PyTorch; compute lives behind the C ABI."""

    def __init__(self, cfg: N.Config):
        super().__init__()
        self._cfg_proto = cfg
        self.compute_dtype: Optional[str] = None  # None -> fp32, or autocast's dtype when active
        object.__setattr__(self, "_handles", {})   # (device index, dtype code) -> (Handle, version signature, source tensors)
        object.__setattr__(self, "_workspaces", {})
        object.__setattr__(self, "_plist_cache", None)
        desc = N.Handle(self._make_cfg(N.LLIE_F32))  # describes the state_dict; validates the topology
        self._param_list = desc.params()
        desc.close()
        params: Dict[str, nn.Parameter] = {}
        for key, shape in self._param_list:
            p = nn.Parameter(torch.empty(shape, dtype=torch.float32))
            _register(self, key, p)
            params[key] = p
        _default_init_(params)

    # -- plumbing -------------------------------------------------------------------------------
    def _make_cfg(self, dtype_code: int) -> N.Config:
        c = N.Config()
        for f, _ in N.Config._fields_:
            setattr(c, f, getattr(self._cfg_proto, f))
        c.compute_dtype = dtype_code
        return c

    # Ordered Parameter objects (llie_param_info order).  What is cached is *where* each one lives -- the `_parameters`
    # dict of its container and the leaf name -- never the Parameter objects: `load_state_dict(assign=True)`, `mod.weight =
    # nn.Parameter(...)`, `register_parameter` and `_apply` with the overwrite-on-conversion future flag all replace the
    # objects inside those dicts, and the engine must follow them (weights to repack, gradients to fill).  321 dict
    # lookups per call instead of a walk over named_parameters() (~1 ms).  The containers themselves are only replaced
    # by assigning a whole sub-module, which the identity check over the ~90 parent links notices.
    def _plist(self):
        cache = self.__dict__.get("_plist_cache")
        if cache is None:
            links, seen, slots = [], set(), []   # links: (parent, name, child) for every container on a parameter's path
            for key, _ in self._param_list:
                mod, parts = self, key.split(".")
                for part in parts[:-1]:
                    child = mod._modules[part]
                    if id(child) not in seen:
                        seen.add(id(child))
                        links.append((mod, part, child))
                    mod = child
                slots.append((mod, parts[-1]))
            cache = (links, slots)
            object.__setattr__(self, "_plist_cache", cache)
        links, slots = cache
        for parent, name, child in links:
            if parent._modules.get(name) is not child:  # a whole sub-module was swapped: re-resolve the paths
                object.__setattr__(self, "_plist_cache", None)
                return self._plist()
        return [mod._parameters[leaf] for mod, leaf in slots]

    def _apply(self, fn, *args, **kwargs):
        object.__setattr__(self, "_plist_cache", None)
        return super()._apply(fn, *args, **kwargs)

    def _signature(self):
        return tuple((p._version, p.data_ptr()) for p in self._plist())

    # copy.deepcopy(model) is how the reference builds its EMA / target networks (low_light_diffusion.py:312,
    # lcm_scheduler.py:353): the copy gets its own engine context lazily, ctypes handles are not copied.
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_handles"], st["_workspaces"], st["_plist_cache"] = {}, {}, None
        return st

    def __setstate__(self, state):
        super().__setstate__(state)
        object.__setattr__(self, "_handles", {})
        object.__setattr__(self, "_workspaces", {})
        object.__setattr__(self, "_plist_cache", None)

    def mark_weights_dirty(self) -> None:
        """Force a full repack at the next call (never needed for correctness: in-place writes that PyTorch's version
        counters miss are caught by the engine's on-device content hash, llie_refresh_params)."""
        for key, (h, _sig, _ts) in list(self._handles.items()):
            self._handles[key] = (h, None, None)

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _handle(self, dtype_code: int) -> N.Handle:
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError(
                "the LCM hot path runs only on a HIP device (got parameters on "
                f"'{dev}'); there is no CPU fallback -- move the model with .to('cuda')")
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), dtype_code)
        sig = self._signature()
        entry = self._handles.get(key)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if entry is not None and entry[1] == sig:
                # same tensors, same version counters: `p.data.copy_()` style writes (the reference's EMA, trainer.py:104-117)
                # are still possible -- the engine compares a content hash on the device and reloads only if it differs
                entry[0].refresh(entry[2], stream)
                return entry[0]
            h = entry[0] if entry is not None else N.Handle(self._make_cfg(dtype_code))
            ts = []
            for t in self._plist():
                t = t.detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                ts.append(t)
            h.load_all(ts, stream)  # one repack launch for everything: an optimiser step touches every tensor
            # a non-fp32 / non-contiguous parameter is loaded from a temporary: no stable source for the hash, reload each time
            stable = all(a.data_ptr() == b.data_ptr() for a, b in zip(ts, self._plist()))
        self._handles[key] = (h, sig if stable else None, ts)
        return h

    def _workspace(self, h: N.Handle, nbytes: int, dev: torch.device) -> torch.Tensor:
        key = (dev.index, id(h))
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            self._workspaces[key] = ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws

    def _wants_grad(self, *tensors) -> bool:
        if not torch.is_grad_enabled():
            return False
        return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in self.parameters())

    def _run_module(self, x: torch.Tensor, temb: Optional[torch.Tensor], out_shape) -> torch.Tensor:
        if x.dim() != 4:
            raise ValueError("expected a [B, C, H, W] tensor")
        if self._wants_grad(x, temb):
            return _ModuleFn.apply(self, out_shape, x, temb, *[p for _, p in self._ordered_params()])
        return self._run_module_raw(x, temb, out_shape)

    def _ordered_params(self):
        return [(k, p) for (k, _), p in zip(self._param_list, self._plist())]

    def _run_module_backward(self, x: torch.Tensor, temb: Optional[torch.Tensor], dy: torch.Tensor):
        h = self._handle(resolve_compute_dtype(self.compute_dtype))
        dev = x.device
        x = x.detach().float().contiguous()
        dy = dy.detach().float().contiguous()
        b, _, hh, ww = x.shape
        nbytes = h.train_workspace_bytes(b, hh, ww)
        ws = self._workspace(h, nbytes, dev)
        dx = torch.empty_like(x)
        flat = torch.empty(h.grad_numel(), dtype=torch.float32, device=dev)
        tp = dtp = None
        dtemb = None
        if temb is not None:
            temb = temb.detach().float().contiguous()
            dtemb = torch.empty_like(temb)
            tp, dtp = temb.data_ptr(), dtemb.data_ptr()
        with torch.cuda.device(dev):
            N.check(N.lib().llie_module_backward(h.h, x.data_ptr(), tp, dy.data_ptr(), dx.data_ptr(), dtp, flat.data_ptr(), b, hh, ww,
                                                 ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream), "backward")
        return dx, dtemb, _grad_views(flat, h, self._param_list)

    def _run_module_raw(self, x: torch.Tensor, temb: Optional[torch.Tensor], out_shape) -> torch.Tensor:
        h = self._handle(resolve_compute_dtype(self.compute_dtype))
        dev = x.device
        x = x.detach().float().contiguous()
        b, _, hh, ww = x.shape
        y = torch.empty((b,) + tuple(out_shape(hh, ww)), dtype=torch.float32, device=dev)
        nbytes = h.workspace_bytes(b, hh, ww)
        ws = self._workspace(h, nbytes, dev)
        tp = None
        if temb is not None:
            temb = temb.detach().float().contiguous()
            tp = temb.data_ptr()
        with torch.cuda.device(dev):
            N.check(N.lib().llie_module_forward(h.h, x.data_ptr(), tp, y.data_ptr(), b, hh, ww, ws.data_ptr(), nbytes,
                                                torch.cuda.current_stream(dev).cuda_stream), "forward")
        return y


def _module_cfg(kind: int, cin: int, cout: int = 0, tdim: int = 0, expansion: int = 4, heads: int = 4,
                split: int = 0) -> N.Config:
    c = N.Config()
    c.kind, c.in_channels, c.out_channels = kind, cin, cout
    c.time_embed_dim, c.expansion_ratio, c.num_attention_heads = tdim, expansion, heads
    c.base_channels = split
    return c


class InvertedResidualBlock(_NativeModule):
    """efficient_unet.py:134-236.  Only the configuration the network uses is built: stride 1, SE on,
    ReLU6 (`quantization_friendly`), dropout 0.  `concat_split` > 0 feeds the block its input as two
    NHWC tensors (channels [0, split) and [split, Cin)), exercising the virtual-concat path the
    decoder uses (efficient_unet.py:588)."""

    def __init__(self, in_channels: int, out_channels: int, time_embed_dim: int, expansion_ratio: int = 4,
                 stride: int = 1, use_se: bool = True, se_ratio: float = 0.25, dropout: float = 0.0,
                 quantization_friendly: bool = True, concat_split: int = 0):
        if stride != 1 or not use_se or se_ratio != 0.25 or dropout != 0.0 or not quantization_friendly:
            raise NotImplementedError("only the configuration instantiated by EfficientUNet is supported")
        super().__init__(_module_cfg(N.LLIE_IRB, in_channels, out_channels, time_embed_dim, expansion_ratio,
                                     split=concat_split))
        self.in_channels, self.out_channels = in_channels, out_channels

    def forward(self, x: torch.Tensor, time_emb: torch.Tensor) -> torch.Tensor:
        return self._run_module(x, time_emb, lambda h, w: (self.out_channels, h, w))


class LinearAttention(_NativeModule):
    """efficient_unet.py:239-308 (dim_head fixed at 32)."""

    def __init__(self, channels: int, num_heads: int = 4, dim_head: int = 32, quantization_friendly: bool = True):
        if dim_head != 32:
            raise NotImplementedError("dim_head is 32 everywhere in the reference network")
        super().__init__(_module_cfg(N.LLIE_ATTN, channels, channels, heads=num_heads))
        self.channels = channels

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_module(x, None, lambda h, w: (self.channels, h, w))


class Downsample(_NativeModule):
    """efficient_unet.py:360-372 (use_conv=True branch: dense 3x3, stride 2, pad 1, bias)."""

    def __init__(self, channels: int, use_conv: bool = True):
        if not use_conv:
            raise NotImplementedError("the AvgPool2d branch is never instantiated by the reference")
        super().__init__(_module_cfg(N.LLIE_DOWN, channels, channels))
        self.channels = channels

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_module(x, None, lambda h, w: (self.channels, h // 2, w // 2))


class Upsample(_NativeModule):
    """efficient_unet.py:375-384 (bilinear x2, align_corners=False, then dense 3x3, bias)."""

    def __init__(self, channels: int):
        super().__init__(_module_cfg(N.LLIE_UP, channels, channels))
        self.channels = channels

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run_module(x, None, lambda h, w: (self.channels, h * 2, w * 2))


class SqueezeExcitation(_NativeModule):
    """efficient_unet.py:79-100 (reduction 0.25, ReLU6 -> sigmoid; `fc1` / `fc2` 1x1 convs with bias).  Inside the network
    SE runs fused into the block's kernels; this module exposes the same kernels for operator-level tests (forward only)."""

    def __init__(self, channels: int, reduction: float = 0.25, quantization_friendly: bool = True):
        if reduction != 0.25 or not quantization_friendly:
            raise NotImplementedError("only the configuration instantiated by EfficientUNet is supported")
        super().__init__(_module_cfg(N.LLIE_SE, channels, channels))
        self.channels = channels

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._wants_grad(x):
            raise NotImplementedError("the stand-alone SqueezeExcitation operator is forward-only; run under torch.no_grad()")
        return self._run_module_raw(x, None, lambda h, w: (self.channels, h, w))


class EfficientUNet(_NativeModule):
    """efficient_unet.py:387-628: same state_dict (321 keys for small@256), forward(x, timestep)."""

    def __init__(self, config: EfficientUNetConfig):
        if not config.use_linear_attention or not config.use_se or not config.quantization_friendly \
                or config.dropout != 0.0 or config.se_ratio != 0.25:
            raise NotImplementedError("only the options used by the reference's variants are supported")
        if len(config.channel_multipliers) != 4 or len(config.attention_resolutions) != 2:
            raise NotImplementedError("4 resolution levels and 2 attention resolutions, like every reference variant")
        c = N.Config()
        c.kind = N.LLIE_UNET
        c.in_channels, c.out_channels, c.base_channels = config.in_channels, config.out_channels, config.base_channels
        for i, m in enumerate(config.channel_multipliers):
            c.channel_multipliers[i] = m
        c.num_res_blocks, c.expansion_ratio = config.num_res_blocks, config.expansion_ratio
        c.time_embed_dim, c.num_attention_heads = config.time_embed_dim, config.num_attention_heads
        c.image_size = config.image_size
        c.attention_resolutions[0], c.attention_resolutions[1] = config.attention_resolutions
        c.allow_unpinned = int(bool(config.allow_unpinned_groupnorm))
        super().__init__(c)
        self.config = config

    # -- native entry points used by the pipeline ----------------------------------------------
    def _prepare(self, batch: int, device: torch.device, enhance_steps: int = 0):
        h = self._handle(resolve_compute_dtype(self.compute_dtype))
        nbytes = h.enhance_workspace_bytes(batch, enhance_steps) if enhance_steps else h.workspace_bytes(batch)
        ws = self._workspace(h, nbytes, device)
        return h, ws, ws.numel()

    @torch.no_grad()
    def time_embedding(self, timestep: torch.Tensor):
        """SinusoidalPosEmb and time_mlp on their own (efficient_unet.py:60-76, 412-417): -> (emb [B, base_channels],
        t_emb [B, time_embed_dim]) as the engine's time kernel computes them (operator-level tests)."""
        h = self._handle(resolve_compute_dtype(self.compute_dtype))
        dev = self._device()
        t = timestep.to(device=dev, dtype=torch.long).contiguous()
        n = t.numel()
        emb = torch.empty(n, self.config.base_channels, dtype=torch.float32, device=dev)
        temb = torch.empty(n, self.config.time_embed_dim, dtype=torch.float32, device=dev)
        stemb = torch.empty_like(temb)
        with torch.cuda.device(dev):
            N.check(N.lib().llie_time_embed(h.h, t.data_ptr(), n, emb.data_ptr(), temb.data_ptr(), stemb.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream), "time_embedding")
        return emb, temb

    def forward_split(self, latents: torch.Tensor, cond: torch.Tensor, timestep: torch.Tensor,
                      uniform_t: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """forward() on the two halves of cat([latents, cond], 1) (low_light_diffusion.py:222) -- the
        concat is never materialised."""
        s = self.config.image_size
        b = latents.shape[0]
        if tuple(latents.shape[2:]) != (s, s) or tuple(cond.shape[2:]) != (s, s):
            raise ValueError(f"this engine runs the UNet at its construction size {s}x{s} "
                             f"(got {tuple(latents.shape[2:])}); attention placement is fixed by image_size")
        if latents.shape[1] + cond.shape[1] != self.config.in_channels or latents.shape[1] != self.config.in_channels // 2:
            raise ValueError("channel split does not match in_channels")
        dev = latents.device
        latents = latents.detach().float().contiguous()
        cond = cond.detach().float().contiguous()
        t = timestep.to(device=dev, dtype=torch.long).contiguous()
        if t.numel() != b:
            raise ValueError("timestep must have one entry per batch row")
        if self._wants_grad() and not uniform_t and out is None:
            # training: activations are kept and the result carries a grad_fn (parameter gradients only)
            return _UNetFn.apply(self, latents, cond, t, *[p for _, p in self._ordered_params()])
        h, ws, nbytes = self._prepare(b, dev)
        if out is None:
            out = torch.empty(b, self.config.out_channels, s, s, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(N.lib().llie_unet_forward(h.h, latents.data_ptr(), cond.data_ptr(), t.data_ptr(), int(uniform_t),
                                              out.data_ptr(), b, ws.data_ptr(), nbytes,
                                              torch.cuda.current_stream(dev).cuda_stream), "EfficientUNet.forward")
        return out

    def forward(self, x: torch.Tensor, timestep: torch.Tensor, return_features: bool = False) -> torch.Tensor:
        if return_features:
            raise NotImplementedError("return_features is a debugging aid of the reference; not provided")
        half = self.config.in_channels // 2
        return self.forward_split(x[:, :half], x[:, half:], timestep)

    # -- reference helpers ------------------------------------------------------------------------
    def get_num_params(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def get_memory_footprint(self, input_size: Tuple[int, int] = (256, 256)) -> dict:
        n = self.get_num_params()  # efficient_unet.py:612-628
        return {"num_params": n, "fp32_mb": n * 4 / (1024 ** 2), "fp16_mb": n * 2 / (1024 ** 2), "int8_mb": n / (1024 ** 2)}


_VARIANTS = {  # efficient_unet.py:646-687
    "tiny": dict(base_channels=16, num_res_blocks=1, expansion_ratio=2, time_embed_dim=64, num_attention_heads=2),
    "small": dict(base_channels=32, num_res_blocks=2, expansion_ratio=4, time_embed_dim=128, num_attention_heads=4),
    "base": dict(base_channels=48, num_res_blocks=2, expansion_ratio=4, time_embed_dim=192, num_attention_heads=6),
    "large": dict(base_channels=64, num_res_blocks=3, expansion_ratio=4, time_embed_dim=256, num_attention_heads=8),
}


def create_efficient_unet(variant: str = "small", image_size: int = 256, **kwargs) -> EfficientUNet:
    """efficient_unet.py:631-692.  `tiny` and `base` raise ValueError at construction exactly like the
    reference (GroupNorm(32, 48)) unless `allow_unpinned_groupnorm=True` is passed (an
    explicitly parity-unpinned deviation, see EfficientUNetConfig)."""
    if variant not in _VARIANTS:
        raise ValueError(f"Unknown variant: {variant}. Choose from {list(_VARIANTS.keys())}")
    cfg = EfficientUNetConfig(channel_multipliers=(1, 2, 4, 8), image_size=image_size, **_VARIANTS[variant], **kwargs)
    return EfficientUNet(cfg)
