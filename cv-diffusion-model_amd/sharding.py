"""Batch sharding across the GPUs of one node: one process per GPU, no data-path collective except a
single all_gather of the fp32 outputs at the end of `enhance`.

Nothing on the path mixes samples (GroupNorm, SE and attention are per sample; no BatchNorm), so each
rank runs `enhance` on a contiguous slice of the batch with replicated weights.  Noise is sliced
from the *global* draw so results do not depend on the world size.  Backend "nccl" is RCCL over xGMI
on ROCm; "gloo" is used by the CPU tests of this module's logic.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Tuple

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


# ------------------------------------------------------------------------------------------------------------
# Training under data parallelism (BASELINE config 5): replicas + one gradient all-reduce per step.
def all_reduce_gradients(params: Iterable[torch.nn.Parameter], *, bucket_bytes: int = 256 << 20, average: bool = True,
                         group=None) -> int:
    """Sum (average) `.grad` of `params` over the ranks with as few, as large collectives as possible; returns
    the number of all_reduce calls issued.

    The engine's backward pass writes every parameter gradient into ONE flat fp32 buffer and hands autograd
    views of it; when the `.grad` tensors still alias a common storage, that storage is reduced in place
    (small: 72 MB -> a single RCCL all-reduce, which is what point-to-point xGMI links want: the ring is
    per-link bound, so few large messages beat many small ones).  Otherwise gradients are packed into flat
    buckets of `bucket_bytes`, reduced, and copied back.  Call between `loss.backward()` and the gradient
    clipping / optimizer step of the reference trainer (trainer.py:300-318).  No-op without a process group."""
    if not dist.is_available() or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    grads: List[torch.Tensor] = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return 0
    calls = 0
    base = grads[0].untyped_storage()
    lo = min(g.storage_offset() for g in grads)
    hi = max(g.storage_offset() + g.numel() for g in grads)
    aliased = (all(g.dtype == torch.float32 and g.is_contiguous() and g.untyped_storage().data_ptr() == base.data_ptr()
                   for g in grads)
               and hi - lo == sum(g.numel() for g in grads))  # one dense span: reducing it reduces every gradient once
    if aliased:
        flat = torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(base, lo, (hi - lo,))
        step = max(1, bucket_bytes // 4)
        for o in range(0, flat.numel(), step):
            dist.all_reduce(flat[o:o + step], group=group)
            calls += 1
        if average:
            flat.div_(world)
        return calls
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size, calls
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1).float() for g in bucket])
        dist.all_reduce(flat, group=group)
        calls += 1
        if average:
            flat.div_(world)
        o = 0
        for g in bucket:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        bucket, size = [], 0

    for g in grads:
        if size and size + g.numel() * 4 > bucket_bytes:
            flush()
        bucket.append(g)
        size += g.numel() * 4
    flush()
    return calls
