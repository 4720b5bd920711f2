"""Runs a few default-schedule forwards at the bench workload and nothing else (for timeline traces).
usage: python tools/fwd_loop.py [steps] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402
from oareactdiff_amd.synthetic import make_inputs, make_topology  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
cfg = dict(PRODUCTION_LEFTNET_CONFIG)
dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                   condition_nf=1, device=dev)
dyn.load_state_dict(synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg), strict=True)
dyn.nan_check = "async"
cm, nfs, ei, masks = make_topology(B, 23)
cm, nfs, ei = cm.to(dev), nfs.to(dev), ei.to(dev)
xh = make_inputs(B, 23, masks, 1, dev)
t = torch.full((B, 1), 0.5, device=dev)
cond = torch.zeros(B, 1, device=dev)
import time
with torch.no_grad():
    for i in range(steps):
        if i == steps // 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        dyn(xh, ei, t, cond, nfs, cm)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 1e3 / (steps - steps // 2))
