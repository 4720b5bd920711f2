"""Find the first stage producing non-finite values for large reactions (no oracle needed).
usage: python tools/nan_hunt.py nf [nf ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from test_hip_parity import _random_case  # noqa: E402
from oareactdiff_amd import _capi  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=3)
sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=7)
L = _capi.lib()
dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                   condition_nf=1, device=dev)
dyn.load_state_dict(sd, strict=True)
dyn.nan_check = "async"
for nf in [int(a) for a in sys.argv[1:]]:
    xh, ei, t, cond, nfs, cm = _random_case([nf] * int(os.environ.get("NB", "1")), float(os.environ.get("PS", "1")), 23, cfg)
    args = ([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))

    def run(stop):
        L.oard_debug_stop_after(stop)
        with torch.no_grad():
            out, _ = dyn(*args)
        torch.cuda.synchronize()
        L.oard_debug_stop_after(0)
        return out

    def fin(tag, which):
        x = dyn.debug_tap(which)
        bad = int((~torch.isfinite(x)).sum())
        print(f"  nf={nf} {tag:16s} shape={tuple(x.shape)} nonfinite={bad} max={float(x[torch.isfinite(x)].abs().max()) if x.numel() else 0:.3e}")

    run(1)
    fin("pos_frame", _capi.TAP_POS_FRAME)
    fin("s0", _capi.TAP_S)
    fin("NE1", _capi.TAP_NE1)
    fin("edge0", _capi.TAP_EDGE)
    for l in range(3):
        run(100 + 10 * l + 1)
        fin(f"l{l}.s_gcl", _capi.TAP_S)
        fin(f"l{l}.edge", _capi.TAP_EDGE)
        run(100 + 10 * l + 2)
        fin(f"l{l}.s", _capi.TAP_S)
    out = run(0)
    print("  final finite:", [bool(torch.isfinite(o).all()) for o in out], "status", dyn.last_status.tolist())
