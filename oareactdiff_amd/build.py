"""Builds liboard_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.environ.get("OARD_LIB") or os.path.join(CSRC, "liboard_hip.so")
SOURCES = ["oard_hip.hip"]


def _headers():
    """Every header the translation unit can include: a stale library must never survive an edit."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "oard.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + _headers())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result"] + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
