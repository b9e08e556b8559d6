"""Functional fp32 CPU restatement of EfficientUNet.forward (efficient_unet.py:532-606).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Works on a flat {state_dict key: tensor} mapping so
that it shares no module code with either the reference or the product.  NCHW like the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .spec import UNetSpec, block_plan, gn_groups

SD = Dict[str, torch.Tensor]


def sinusoidal_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """efficient_unet.py:68-76 -- [cos | sin], frequencies exp(-ln(P) * i / half), always fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    c = x.shape[1]
    # gn_groups(c) == min(32, c) for every channel count the reference can construct; see spec.gn_groups
    return F.group_norm(x, gn_groups(c), sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def time_embed(sd: SD, spec: UNetSpec, t: torch.Tensor, pre: str) -> torch.Tensor:
    """efficient_unet.py:412-417."""
    e = sinusoidal_embedding(t, spec.base_channels)
    e = F.linear(e, sd[pre + "time_mlp.1.weight"], sd[pre + "time_mlp.1.bias"])
    return F.linear(F.silu(e), sd[pre + "time_mlp.3.weight"], sd[pre + "time_mlp.3.bias"])


def se_gate(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """efficient_unet.py:96-100 -- returns the per-(sample, channel) gate in (0,1), shape [B,C,1,1]."""
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.conv2d(s, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"]).clamp(0.0, 6.0)
    return torch.sigmoid(F.conv2d(s, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"]))


def irb_forward(sd: SD, p: str, x: torch.Tensor, temb: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """InvertedResidualBlock.forward, efficient_unet.py:203-236 (ReLU6 variant, dropout p=0)."""
    hid = sd[p + ".expand.weight"].shape[0]
    h = _gn(sd, p + ".norm1", x).clamp(0.0, 6.0)
    h = F.conv2d(h, sd[p + ".expand.weight"])
    if taps is not None:
        taps["expand"] = h
    h = _gn(sd, p + ".norm2", h)
    film = F.linear(F.silu(temb), sd[p + ".time_mlp.1.weight"], sd[p + ".time_mlp.1.bias"])
    scale, shift = film[:, :hid, None, None], film[:, hid:, None, None]
    h = (h * (1 + scale) + shift).clamp(0.0, 6.0)
    h = F.conv2d(h, sd[p + ".depthwise.weight"], padding=1, groups=hid)
    if taps is not None:
        taps["depthwise"] = h
    h = h * se_gate(sd, p + ".se", h)
    h = F.conv2d(h, sd[p + ".project.weight"])
    if (p + ".skip.weight") in sd:
        return h + F.conv2d(x, sd[p + ".skip.weight"])
    return h + x  # Cin == Cout, stride 1 everywhere in this network (efficient_unet.py:164)


def linear_attention_forward(sd: SD, p: str, x: torch.Tensor, heads: int, dim_head: int = 32) -> torch.Tensor:
    """LinearAttention.forward, efficient_unet.py:273-308.  phi = elu+1 on q,k; no `scale`."""
    b, c, hh, ww = x.shape
    n = hh * ww
    qkv = F.conv2d(_gn(sd, p + ".norm", x), sd[p + ".to_qkv.weight"])
    q, k, v = (z.reshape(b, heads, dim_head, n) for z in qkv.chunk(3, dim=1))  # head-major channels
    q = F.elu(q) + 1
    k = F.elu(k) + 1
    ksum = k.sum(dim=-1)                                   # [b,h,d]
    kv = torch.einsum("bhdn,bhen->bhde", k, v)             # [b,h,d,e]
    num = torch.einsum("bhdn,bhde->bhen", q, kv)           # [b,h,e,n]
    den = torch.einsum("bhdn,bhd->bhn", q, ksum)[:, :, None, :] + 1e-6
    out = (num / den).reshape(b, heads * dim_head, hh, ww)
    out = F.conv2d(out, sd[p + ".to_out.0.weight"])
    return _gn(sd, p + ".to_out.1", out) + x


def downsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """efficient_unet.py:367 -- dense 3x3, stride 2, pad 1, bias."""
    return F.conv2d(x, sd[p + ".down.weight"], sd[p + ".down.bias"], stride=2, padding=1)


def upsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """efficient_unet.py:383-384 -- bilinear x2 (align_corners=False) then dense 3x3 pad 1, bias."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    return F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)


def unet_forward(sd: SD, spec: UNetSpec, x: torch.Tensor, t: torch.Tensor, prefix: str = "unet.",
                 trace: Optional[List] = None) -> torch.Tensor:
    """EfficientUNet.forward, efficient_unet.py:532-606.  `trace`, if given, receives
    (name, tensor) after every block for layer-wise debugging of the HIP path."""
    plan = block_plan(spec)
    pre = prefix
    heads = spec.num_attention_heads

    def run(blocks, h):
        for b in blocks:
            if b[0] == "irb":
                h = irb_forward(sd, pre + b[1], h, temb)
            else:
                h = linear_attention_forward(sd, pre + b[1], h, heads, spec.dim_head)
            if trace is not None:
                trace.append((b[1], h))
        return h

    temb = time_embed(sd, spec, t, pre)
    h = F.conv2d(x, sd[pre + "init_conv.weight"], sd[pre + "init_conv.bias"], padding=1)
    if trace is not None:
        trace.append(("init_conv", h))
    skips = []
    n_lvl = len(plan["enc"])
    for lvl, blocks in enumerate(plan["enc"]):
        h = run(blocks, h)
        skips.append(h)                                   # one skip per level, before the downsample (:567)
        if lvl < n_lvl - 1:
            h = downsample(sd, pre + f"downsamplers.{lvl}", h)
            if trace is not None:
                trace.append((f"downsamplers.{lvl}", h))
    h = run(plan["mid"], h)
    for lvl, blocks in enumerate(plan["dec"]):
        if lvl > 0:
            h = upsample(sd, pre + f"upsamplers.{lvl - 1}", h)
            if trace is not None:
                trace.append((f"upsamplers.{lvl - 1}", h))
        h = torch.cat([h, skips.pop()], dim=1)            # h first, skip second (:588)
        h = run(blocks, h)
    h = F.silu(_gn(sd, pre + "final_norm", h))
    return F.conv2d(h, sd[pre + "final_conv.weight"], sd[pre + "final_conv.bias"], padding=1)
