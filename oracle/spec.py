"""Topology of the denoiser: variant table, block plan and state_dict key/shape grammar.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows:
  * variant table            efficient_unet.py:646-687
  * encoder/decoder builder  efficient_unet.py:403-530 (attention placement :426,447,463,509,525)
  * IRB parameter set        efficient_unet.py:147-201, SE :85-94, LinearAttention :250-271
  * wrapper in_channels=6    low_light_diffusion.py:77
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class UNetSpec:
    in_channels: int = 6
    out_channels: int = 3
    base_channels: int = 32
    channel_multipliers: Tuple[int, ...] = (1, 2, 4, 8)
    attention_resolutions: Tuple[int, ...] = (16, 8)
    num_attention_heads: int = 4
    num_res_blocks: int = 2
    expansion_ratio: int = 4
    se_ratio: float = 0.25
    time_embed_dim: int = 128
    image_size: int = 256
    dim_head: int = 32  # LinearAttention default, efficient_unet.py:254
    # tiny / base cannot be constructed by the reference (GroupNorm(32, 48)).  With allow_unpinned the oracle builds
    # them with groups = gn_groups(C) below; nothing pins that choice (parity-unpinned), it only gives the engine's
    # opt-in tiny / base something to be compared with.
    allow_unpinned: bool = False

    @property
    def channels(self) -> List[int]:
        return [self.base_channels * m for m in self.channel_multipliers]


# efficient_unet.py:646-687
VARIANTS: Dict[str, dict] = {
    "tiny": dict(base_channels=16, num_res_blocks=1, expansion_ratio=2, time_embed_dim=64, num_attention_heads=2),
    "small": dict(base_channels=32, num_res_blocks=2, expansion_ratio=4, time_embed_dim=128, num_attention_heads=4),
    "base": dict(base_channels=48, num_res_blocks=2, expansion_ratio=4, time_embed_dim=192, num_attention_heads=6),
    "large": dict(base_channels=64, num_res_blocks=3, expansion_ratio=4, time_embed_dim=256, num_attention_heads=8),
}


def make_spec(variant: str = "small", image_size: int = 256, in_channels: int = 6, **kw) -> UNetSpec:
    if variant not in VARIANTS:  # efficient_unet.py:689-690
        raise ValueError(f"Unknown variant: {variant}. Choose from {list(VARIANTS.keys())}")
    return UNetSpec(in_channels=in_channels, image_size=image_size, **VARIANTS[variant], **kw)


def gn_groups(c: int) -> int:
    """min(32, C) wherever nn.GroupNorm accepts it (every constructible variant); otherwise the largest divisor of C
    that is <= 32 (48 -> 24, 144 -> 24): the unpinned deviation shared with the engine (engine.cpp: gn_groups)."""
    g = min(32, c)
    while c % g:
        g -= 1
    return g


def _check_gn(c: int, allow_unpinned: bool = False) -> None:
    g = min(32, c)
    if c % g != 0 and not allow_unpinned:  # what nn.GroupNorm(min(32, C), C) raises at efficient_unet.py:170 for tiny/base
        raise ValueError("num_channels must be divisible by num_groups")


def block_plan(spec: UNetSpec) -> dict:
    """Ordered description of the module tree.  Each block is ("irb", prefix, cin, cout) or
    ("attn", prefix, c).  Attention placement is decided from the *tracked* resolution that starts at
    spec.image_size (efficient_unet.py:426,447,509), not from the runtime tensor."""
    ch = spec.channels
    res = spec.image_size
    enc, dec = [], []
    in_ch = ch[0]
    for lvl, out_ch in enumerate(ch):
        blocks, k = [], 0
        for b in range(spec.num_res_blocks):
            blocks.append(("irb", f"encoder_blocks.{lvl}.{k}", in_ch if b == 0 else out_ch, out_ch)); k += 1
            if res in spec.attention_resolutions:
                blocks.append(("attn", f"encoder_blocks.{lvl}.{k}", out_ch)); k += 1
        enc.append(blocks)
        in_ch = out_ch
        if lvl < len(ch) - 1:
            res //= 2
    mid = [("irb", "mid_block1", ch[-1], ch[-1]), ("attn", "mid_attn", ch[-1]), ("irb", "mid_block2", ch[-1], ch[-1])]
    rev = list(reversed(ch))
    for lvl, out_ch in enumerate(rev):
        blocks, k = [], 0
        for b in range(spec.num_res_blocks + 1):
            cin = in_ch + out_ch if b == 0 else out_ch
            blocks.append(("irb", f"decoder_blocks.{lvl}.{k}", cin, out_ch)); k += 1
            if res in spec.attention_resolutions:
                blocks.append(("attn", f"decoder_blocks.{lvl}.{k}", out_ch)); k += 1
        dec.append(blocks)
        in_ch = out_ch
        if lvl < len(rev) - 1:
            res *= 2
    return dict(enc=enc, mid=mid, dec=dec, down=ch[:-1], up=rev[:-1], channels=ch)


def param_shapes(spec: UNetSpec, prefix: str = "unet.") -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape, in the reference's registration order (SURVEY.md 8b grammar)."""
    T, e = spec.time_embed_dim, spec.expansion_ratio
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def put(k, *shape):
        out[prefix + k] = tuple(shape)

    def gn(k, c):
        _check_gn(c, spec.allow_unpinned)
        put(k + ".weight", c); put(k + ".bias", c)

    def irb(p, cin, cout):
        hid = int(cin * e)
        sq = max(1, int(hid * spec.se_ratio))
        gn(p + ".norm1", cin); gn(p + ".norm2", hid)
        put(p + ".expand.weight", hid, cin, 1, 1)
        put(p + ".depthwise.weight", hid, 1, 3, 3)
        put(p + ".se.fc1.weight", sq, hid, 1, 1); put(p + ".se.fc1.bias", sq)
        put(p + ".se.fc2.weight", hid, sq, 1, 1); put(p + ".se.fc2.bias", hid)
        put(p + ".project.weight", cout, hid, 1, 1)
        put(p + ".time_mlp.1.weight", 2 * hid, T); put(p + ".time_mlp.1.bias", 2 * hid)
        if cin != cout:
            put(p + ".skip.weight", cout, cin, 1, 1)

    def attn(p, c):
        inner = spec.num_attention_heads * spec.dim_head
        gn(p + ".norm", c)
        put(p + ".to_qkv.weight", 3 * inner, c, 1, 1)
        put(p + ".to_out.0.weight", c, inner, 1, 1)
        gn(p + ".to_out.1", c)

    def blocks(bl):
        for b in bl:
            irb(*b[1:]) if b[0] == "irb" else attn(*b[1:])

    plan = block_plan(spec)
    ch = plan["channels"]
    put("time_mlp.1.weight", T, spec.base_channels); put("time_mlp.1.bias", T)
    put("time_mlp.3.weight", T, T); put("time_mlp.3.bias", T)
    put("init_conv.weight", ch[0], spec.in_channels, 3, 3); put("init_conv.bias", ch[0])
    for lvl in plan["enc"]:
        blocks(lvl)
    for i, c in enumerate(plan["down"]):
        put(f"downsamplers.{i}.down.weight", c, c, 3, 3); put(f"downsamplers.{i}.down.bias", c)
    blocks(plan["mid"])
    for lvl in plan["dec"]:
        blocks(lvl)
    for i, c in enumerate(plan["up"]):
        put(f"upsamplers.{i}.conv.weight", c, c, 3, 3); put(f"upsamplers.{i}.conv.bias", c)
    gn("final_norm", ch[0])
    put("final_conv.weight", spec.out_channels, ch[0], 3, 3); put("final_conv.bias", spec.out_channels)
    return out
