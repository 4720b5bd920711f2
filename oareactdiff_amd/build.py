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


DEFAULT_DIMS = "196x96,32x8,32x32"


def dims_define(spec: str) -> str:
    """OARD_DIMS="196x96,32x8" -> -DOARD_DIMS_LIST=X(196,96)X(32,8): the (hidden_channels, num_radial) pairs to instantiate."""
    pairs = []
    for item in spec.replace(" ", "").split(","):
        if not item:
            continue
        h, r = item.lower().split("x")
        h, r = int(h), int(r)
        if h % 4 or r % 4 or h < 16 or r < 4:
            raise ValueError(f"OARD_DIMS entry {item!r}: widths must be multiples of 4 (hidden >= 16)")
        if (h, r) not in pairs:
            pairs.append((h, r))
    if not pairs:
        raise ValueError("OARD_DIMS is empty")
    return "-DOARD_DIMS_LIST=" + "".join(f"X({h},{r})" for h, r in pairs)


def _dims_file() -> str:
    return LIB + ".dims"


def _built_dims() -> str:
    """The OARD_DIMS the library on disk was compiled with (side file written next to it; absent = unknown -> rebuild)."""
    try:
        with open(_dims_file()) as f:
            return f.read().strip()
    except OSError:
        return ""


def build(force: bool = False, verbose: bool = False) -> str:
    dims = os.environ.get("OARD_DIMS", DEFAULT_DIMS)
    dims_define(dims)                                    # validate before deciding anything
    # a library built with other widths is stale whatever its timestamp says (a non-default OARD_DIMS build used to leave a
    # liboard_hip.so without 196x96 that the next default build() kept)
    if not force and not _stale() and _built_dims() == dims:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", dims_define(dims)] + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, cwd=CSRC, check=True)
    with open(_dims_file(), "w") as f:
        f.write(dims + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
