"""Builds liboard_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.environ.get("OARD_LIB") or os.path.join(CSRC, "liboard_hip.so")
# translation units -> the headers each one includes (None = every header of csrc/ except the ones another unit owns).  Units are compiled
# to objects side by side and linked; an edit recompiles only the units that include the edited file.
_EDGE = ["oard_engine.h", "oard_layout.h", "oard_kernels.h", "oard_edge_v1.h", "oard_inst.h"]
SOURCES = {
    "oard_hip.hip": None,                                                        # host code + every kernel not listed below
    "oard_general.hip": ["oard_general.h"],                                       # general-edge-list path
    # explicit instantiations of the heavy kernel families (csrc/oard_inst.h)
    "oard_inst_gcl_p.hip": _EDGE + ["oard_edge_p.h"],
    "oard_inst_gcl_v1.hip": _EDGE + ["oard_node_v1.h", "oard_edge_small.h"],
    "oard_inst_b3.hip": _EDGE + ["oard_edge_b3.h"],
    "oard_inst_equi.hip": _EDGE + ["oard_node_v1.h", "oard_edge_bwd.h"],
    "oard_inst_wgrad.hip": _EDGE + ["oard_edge_bwd.h", "oard_wgrad_t16.h"],
    "oard_inst_node.hip": _EDGE + ["oard_node_v1.h", "oard_edge_bwd.h", "oard_node_bwd.h", "oard_rows.h"],
}
_OWNED = {"oard_general.h"}                     # headers no other unit includes
_PUBLIC = os.path.join("..", "..", "include", "oard.h")


def _deps(src):
    """Every file whose edit makes `src`'s object stale: a stale library must never survive an edit."""
    own = SOURCES[src]
    hdrs = sorted(f for f in os.listdir(CSRC) if f.endswith(".h") and f not in _OWNED) if own is None else list(own)
    return [src] + hdrs + [_PUBLIC]


def _objdir():
    """Objects are kept per output library: an OARD_LIB build with other widths / flags must not share objects with the default one."""
    return os.path.join(os.path.dirname(os.path.abspath(LIB)), "build", os.path.basename(LIB))


def _obj(src):
    return os.path.join(_objdir(), os.path.splitext(src)[0] + ".o")


def _newer(paths, than):
    if not os.path.exists(than):
        return True
    t = os.path.getmtime(than)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in paths)


def _stale() -> bool:
    return any(_newer(_deps(src), LIB) for src in SOURCES)


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
    os.makedirs(_objdir(), exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", dims_define(dims)]
    extra = os.environ.get("OARD_CXXFLAGS", "").split()          # experiment / ablation builds (-DOARD_EXPERIMENTS ...): always with OARD_LIB
    tag = os.path.join(_objdir(), "flags.txt")
    flag_line = " ".join(flags + extra)
    try:
        same_flags = open(tag).read() == flag_line
    except OSError:
        same_flags = False
    todo = [src for src in SOURCES if force or not same_flags or _newer(_deps(src), _obj(src))]

    def compile_one(src):
        cmd = [hipcc] + flags + extra + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, cwd=CSRC, check=True)
    if todo:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(todo)) as pool:
            list(pool.map(compile_one, todo))
        with open(tag, "w") as f:
            f.write(flag_line)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(src) for src in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, cwd=CSRC, check=True)
    with open(_dims_file(), "w") as f:
        f.write(dims + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
