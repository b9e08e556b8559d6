"""Deterministic, framework-independent synthetic weights keyed by (state_dict key, shape).

TEST INFRASTRUCTURE.  No trained checkpoint ships with the reference and default
init of 18 M parameters is too large to commit, so goldens and parity tests fill the reference, the
oracle and the HIP engine from this generator.  Values depend only on
(seed, key, flat index): a splitmix64 counter hash mapped to a uniform.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

_M64 = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n doubles in [0,1), a pure function of (seed, key, index)."""
    base = (zlib.crc32(key.encode("utf-8")) * 0x100000001B3 + seed * 0xD6E8FEB86659FD93) & _M64
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(base)
    z = _splitmix64(_splitmix64(ctr))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape))
    u = uniform01(key, n, seed) * 2.0 - 1.0  # [-1, 1)
    if len(shape) == 1:
        is_gn_weight = key.endswith(".weight")  # the only 1-D ".weight" tensors are GroupNorm gains
        v = 1.0 + 0.2 * u if is_gn_weight else 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:]))
        v = u * np.sqrt(3.0 / fan_in)  # unit-gain uniform
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, synth_tensor(k, s, seed)) for k, s in shapes.items())
