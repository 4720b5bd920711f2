"""A checkpoint with other widths than the default build instantiates is a REBUILD (`OARD_DIMS`), not a dead end: the
library is compiled into a scratch file with an extra (hidden_channels, num_radial) pair and a forward at those widths
is compared with the float64 oracle - in a child process, because a process binds one library (`OARD_LIB`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/oracle"); sys.path.insert(0, ROOT + "/tests")
import leftnet_oracle as oracle
from oareactdiff_amd import _capi
from oareactdiff_amd.dynamics import EGNNDynamics
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
from oareactdiff_amd.synthetic import make_inputs, make_topology
from _cases import rel
dev = torch.device("cuda:0")
cfg = dict(PRODUCTION_LEFTNET_CONFIG, hidden_channels=64, num_radial=16, num_layers=3)
sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)
dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
dyn.load_state_dict(sd, strict=True)
B, nf = 3, 7
cm, nfs, ei, masks = make_topology(B, nf)
xh = make_inputs(B, nf, masks, 5, "cpu")
t, cond = torch.full((B, 1), 0.4), torch.zeros(B, 1)
with torch.no_grad():
    out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
ref = oracle.dynamics_forward({k: v.double() for k, v in sd.items()}, cfg, [x.double() for x in xh], ei, t.double(), cond.double(),
                              nfs, cm, 1, nodeframe="exact")
ev = max(rel(o[:, :3].cpu(), r[:, :3]) for o, r in zip(out, ref))
eh = max(rel(o[:, 3:].cpu(), r[:, 3:]) for o, r in zip(out, ref))
print(f"DIMS64x16 vel {ev:.3e} h {eh:.3e}")
assert ev <= 1e-5 and eh <= 1e-5
# the widths this scratch build leaves out: inference runs the general-edge-list kernels (run-time widths; a warning names the rebuild),
# training - which exists for built widths only - is refused loudly
import warnings
cfg2 = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2)
sd2 = synthetic_state_dict(state_spec(cfg2, [9, 9, 9], 1), cfg2)
d2 = EGNNDynamics(model_config=dict(cfg2), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
d2.load_state_dict(sd2, strict=True)
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    with torch.no_grad():
        out2, _ = d2([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
assert any("not a width pair the production kernels were built for" in str(w.message) for w in caught), [str(w.message) for w in caught]
assert d2._last_topo.graph is not None and d2._last_topo.handle is None
ref2 = oracle.dynamics_forward({k: v.double() for k, v in sd2.items()}, cfg2, [x.double() for x in xh], ei, t.double(), cond.double(),
                               nfs, cm, 1, nodeframe="literal")
ev2 = max(rel(o[:, :3].cpu(), r[:, :3]) for o, r in zip(out2, ref2))
assert ev2 <= 1e-5, ev2
print(f"GENERAL196x96 vel {ev2:.3e}")
try:
    d2([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))          # under autograd
    raise SystemExit("196x96 is not in this build: its training-mode forward must raise")
except _capi.OardError:
    print("REFUSEDTRAIN196x96")
'''


def test_dims_define():
    from oareactdiff_amd.build import dims_define
    assert dims_define("196x96, 32x8,32x8") == "-DOARD_DIMS_LIST=X(196,96)X(32,8)"
    for bad in ("", "30x8", "196"):
        with pytest.raises(ValueError):
            dims_define(bad)


def test_library_built_with_other_widths_is_stale(tmp_path, monkeypatch):
    """Round-3 advisor finding: a non-default OARD_DIMS build overwrote liboard_hip.so and the next default build() kept it (newer
    than the sources).  The widths a library was built with are recorded next to it; a mismatch means rebuild."""
    from oareactdiff_amd import build as B
    lib = os.path.join(tmp_path, "liboard_x.so")
    calls = []

    def fake_run(cmd, **kw):                                 # a build = one compile per translation unit, then ONE link: count the links
        if "-shared" in cmd:
            calls.append(cmd)
            open(lib, "w").close()
        else:
            compiles.append(cmd)
            open(cmd[cmd.index("-o") + 1], "w").close()
    compiles = []
    monkeypatch.setattr(B, "LIB", lib)
    monkeypatch.setattr(B.subprocess, "run", fake_run)
    monkeypatch.setenv("OARD_DIMS", "64x16")
    B.build()
    assert len(calls) == 1 and B._built_dims() == "64x16"
    B.build()
    assert len(calls) == 1                                   # same widths, library newer than the sources: kept
    monkeypatch.delenv("OARD_DIMS")
    B.build()
    assert len(calls) == 2 and B._built_dims() == B.DEFAULT_DIMS and any("X(196,96)" in a for a in compiles[-1])
    assert len(compiles) == 2 * len(B.SOURCES)               # other widths = other flags: every unit recompiled
    os.remove(lib + ".dims")
    B.build()
    assert len(calls) == 3                                   # unknown widths: rebuilt


@pytest.mark.gpu
def test_rebuild_with_another_width_matches_the_oracle(tmp_path):
    lib = os.path.join(tmp_path, "liboard_dims.so")
    env = dict(os.environ, OARD_DIMS="64x16", OARD_LIB=lib)
    b = subprocess.run([sys.executable, "-m", "oareactdiff_amd.build", "--force"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert b.returncode == 0 and os.path.exists(lib), b.stderr[-2000:]
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "DIMS64x16" in out.stdout and "GENERAL196x96" in out.stdout and "REFUSEDTRAIN196x96" in out.stdout
