"""Builds libllie_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path() -> str:
    return os.path.join(_HERE, "libllie_hip.so")


def build_library(force: bool = False, jobs: int = 8, verbose: bool = False) -> str:
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.run(["make", "-C", csrc, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(["make", "-C", csrc, f"-j{jobs}"], capture_output=True, text=True)
    if verbose:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libllie_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return library_path()
