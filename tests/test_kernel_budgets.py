"""Register budgets of the hot kernels, checked at build time (no GPU): a spill in a production instantiation is a silent
performance regression that no parity test sees (round 4: a run-time `reflect_equiv` flag cost k_scalarize_bwd 69 spilled registers and
the training step 2.7 ms until it became a template parameter).  Compiles the production widths only and reads hipcc's
-Rpass-analysis=kernel-resource-usage remarks."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oareactdiff_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

#: demangled-name regex -> max scratch bytes per lane (the one known exception: the large-batch k_equi_node_v1)
BUDGETS = {
    r"^void k_gcl_edge_v1<Dims<196, 96>, 8, 2, .*, 2, 3>": 0,
    r"^void k_gcl_edge_p<Dims<196, 96>, ": 0,                 # persistent form (round 5): inference and training-mode instantiations
    r"^void k_equi_edge_v1<Dims<196, 96>, 8, ": 0,
    r"^void k_gcl_edge_bwd<": 0,
    r"^void k_equi_edge_bwd<": 0,
    r"^void k_wgrad_t16<": 0,
    r"^void k_wgrad_q<": 0,
    r"^void k_scalarize_bwd<Dims<196, 96>, \d+, (true|false)>": 0,
    r"^void k_equi_msg_bwd<Dims<196, 96> >": 0,
    r"^void k_rows_dense2<": 0,
    r"^void k_rows_dense_long<": 0,
    r"^void k_gcl_edge_b3<": 0,
    r"^void k_equi_edge_b3<Dims<196, 96>, false>": 0,
    r"^void k_equi_edge_b3<Dims<196, 96>, true>": 20,        # the optional split-precision TRAINING-mode forward (tape stores)
    r"^void k_gcl_node_v1<Dims<196, 96>, 13, ": 0,
    r"^void k_equi_node_v1<Dims<196, 96>, 13, (true|false), false>": 36,
}


def _compile_unit(args):
    src, out = args
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-Wno-unused-result",
                        "-DOARD_DIMS_LIST=X(196,96)", "-Rpass-analysis=kernel-resource-usage", src, "-o", out], cwd=CSRC, capture_output=True,
                       text=True, timeout=900)
    return src, r.returncode, r.stderr


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_production_kernels_stay_inside_their_register_budgets(tmp_path):
    """Every translation unit of the library (oareactdiff_amd/build.py SOURCES), production widths, device code only, side by side."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(os.path.dirname(CSRC), ".."))
    from oareactdiff_amd.build import SOURCES
    units = [(src, os.path.join(tmp_path, os.path.splitext(src)[0] + ".o")) for src in SOURCES]
    with ThreadPoolExecutor(max_workers=len(units)) as pool:
        results = list(pool.map(_compile_unit, units))
    blocks = []
    for src, rc, err in results:
        assert rc == 0, (src, err[-2000:])
        blocks += err.split("Function Name: ")[1:]
    names = subprocess.run(["c++filt"], input="\n".join(b.split("\n")[0].split()[0] for b in blocks), capture_output=True,
                           text=True).stdout.splitlines()
    assert len(set(names)) == len(names), "a kernel is emitted by two translation units: " + str(sorted({n for n in names if names.count(n) > 1})[:5])
    seen = {k: 0 for k in BUDGETS}
    bad = []
    for dem, b in zip(names, blocks):
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgpr = int(re.search(r"VGPRs: (\d+)", b).group(1))
        for pat, limit in BUDGETS.items():
            if re.search(pat, dem):
                seen[pat] += 1
                if scratch > limit:
                    bad.append(f"{dem[:140]}: {scratch} bytes of scratch per lane (budget {limit}), {vgpr} VGPRs")
    assert not bad, "\n".join(bad)
    assert all(v > 0 for v in seen.values()), [k for k, v in seen.items() if v == 0]
    # ---- wait states behind MFMA results, in the ISA of the same objects (round 6, advisor finding) --------------------------------------
    # Some kernels issue their MFMAs as `asm volatile` (dense_seq_xyz of csrc/oard_node_v1.h, k_wgrad_t16): hipcc's hazard recogniser does
    # not look inside asm statements, and one of those kernels (k_equi_node_v1) is allowed 36 bytes of scratch - a spill store of an
    # accumulator right behind its last MFMA would read a result that is not there yet.  tools/mfma_hazard_check.py walks the disassembly
    # of EVERY kernel: no instruction but an accumulating MFMA touches an MFMA's destination inside its window.
    sys.path.insert(0, os.path.join(os.path.dirname(CSRC), "..", "tools"))
    import mfma_hazard_check as hz
    n_mfma = n_fn = 0
    viol = []
    for src, out in units:
        co = out[:-2] + ".co"
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + out,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        v, m, f = hz.check(co)
        viol += v
        n_mfma += m
        n_fn += f
    assert n_mfma > 10000 and n_fn > 100, (n_mfma, n_fn)        # the scan saw the library
    assert not viol, "\n".join(viol[:20])


def test_the_hazard_scan_flags_what_it_should():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(CSRC), "..", "tools"))
    import mfma_hazard_check as hz
    mf = "v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]"
    assert hz.scan([mf, "v_mfma_f32_16x16x4_f32 v[0:3], v6, v7, v[0:3]", "s_nop 7", "s_nop 1", "v_add_f32 v8, v0, v1"])[0] == []      # chain, 10 states
    assert hz.scan([mf, "s_nop 7", "s_nop 0", "v_add_f32 v8, v0, v1"])[0]                       # VALU read after 9 states
    assert hz.scan([mf, "scratch_store_dwordx4 off, v[0:3], s0 offset:16"])[0]                   # a spill store right behind the MFMA
    assert hz.scan([mf, "v_mov_b32 v2, 0"])[0]                                                   # a write into the destination
    assert hz.scan([mf, "v_mfma_f32_16x16x4_f32 v[8:11], v0, v5, v[8:11]"])[0]                   # the result as SrcA of the next MFMA
    assert hz.scan([mf, "v_mfma_f32_16x16x4_f32 v[8:11], v6, v5, v[0:3]"])[0] == []              # ... taken whole as SrcC: the chain
    assert hz.scan(["v_mfma_f32_4x4x1_16b_f32 v[0:3], v4, v5, v[0:3]", "s_nop 3", "v_add_f32 v8, v0, v1"])[0] == []      # 2 passes: 4 states
