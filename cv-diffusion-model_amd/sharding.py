"""Batch sharding across the GPUs of one node: one process per GPU, no data-path collective except a
single all_gather of the fp32 outputs at the end of `enhance`.

Nothing on the path mixes samples (GroupNorm, SE and attention are per sample; no BatchNorm), so each
rank runs `enhance` on a contiguous slice of the batch with replicated weights.  Noise is sliced
from the *global* draw so results do not depend on the world size.  Backend "nccl" is RCCL over xGMI
on ROCm; "gloo" is used by the CPU tests of this module's logic.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` items owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_batch(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Concatenate per-rank batch slices (sizes from shard_range) into the full batch on every rank.
    Equal shards use one all_gather_into_tensor (a single RCCL collective); ragged shards pad to the
    largest shard first."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], "local shard does not match shard_range"
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: sizes[rank]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def enhance_sharded(enhance_fn: Callable[..., torch.Tensor], low_light: torch.Tensor, *,
                    noise: Optional[torch.Tensor] = None, gather: bool = True, group=None, **kw) -> torch.Tensor:
    """Run `enhance_fn(low_light[lo:hi], noise=noise[:, lo:hi], **kw)` on this rank's slice of the
    global batch and all_gather the outputs.  `low_light` (and `noise` [steps,B,...] if given) are the
    GLOBAL tensors, identical on every rank (e.g. seeded host draws), so world_size 1 and N agree."""
    if not dist.is_available() or not dist.is_initialized():
        return enhance_fn(low_light, noise=noise, **kw) if noise is not None else enhance_fn(low_light, **kw)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = low_light.shape[0]
    lo, hi = shard_range(total, rank, world)
    args = dict(kw)
    if noise is not None:
        args["noise"] = noise[:, lo:hi]
    local = enhance_fn(low_light[lo:hi], **args)
    return all_gather_batch(local, total, group) if gather else local
