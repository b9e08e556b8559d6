"""Builds libllie_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path() -> str:
    return os.path.join(_HERE, "libllie_hip.so")


def build_library(force: bool = False, jobs: int = 8, verbose: bool = False) -> str:
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, f"-j{jobs}"]
    if force:
        # from scratch, but into a scratch object directory and a scratch target: the shipped in-tree library is replaced
        # only by a build that succeeded (a hipcc failure on a host without the toolchain must not remove it)
        tmp_obj, tmp_so = "build_force", os.path.join(_HERE, "libllie_hip.so.new")
        shutil.rmtree(os.path.join(csrc, tmp_obj), ignore_errors=True)
        cmd += [f"OBJDIR={tmp_obj}", f"TARGET={tmp_so}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose:
        print(r.stdout)
    if force:
        shutil.rmtree(os.path.join(csrc, "build_force"), ignore_errors=True)
    if r.returncode != 0:
        if force and os.path.exists(tmp_so):
            os.remove(tmp_so)
        raise RuntimeError("building libllie_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if force:
        os.replace(tmp_so, library_path())
    return library_path()
