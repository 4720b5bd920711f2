"""Stage timestamps of one k_equi_node_v1 workgroup in an experiment build (-DOARD_EXPERIMENTS -DOARD_TIMELINE):
    OARD_LIB=.../liboard_tl.so PROBE_B=64 python tools/node_timeline.py
codes: 1 entry, 2 gather done, 3 after barrier, 4 vec_proj + frame scalar done, 5 after barrier, 6 xvec hidden done, 7 after barrier, 0 end"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oareactdiff_amd import _capi
from oareactdiff_amd.dynamics import EGNNDynamics
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
from oareactdiff_amd.synthetic import make_inputs, make_topology

B = int(os.environ.get("PROBE_B", "64"))
L = _capi.lib()
for k, v in dict(parts=1, sequential=1).items():
    assert L.oard_debug_option(k.encode(), v) == 0
dev = torch.device("cuda:0")
cfg = dict(PRODUCTION_LEFTNET_CONFIG)
dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
dyn.load_state_dict(synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg), strict=True)
dyn.nan_check = "async"
cm, nfs, ei, masks = make_topology(B, 23)
cm, nfs, ei = cm.to(dev), nfs.to(dev), ei.to(dev)
inp = make_inputs(B, 23, masks, 1234, dev)
cond = torch.zeros(B, 1, device=dev)
t = torch.full((B, 1), 0.5, device=dev)
for _ in range(3):
    with torch.no_grad():
        dyn(inp, ei, t, cond, nfs, cm)
W, TL_MAX = 13, 1024
buf = np.zeros((16, TL_MAX), dtype=np.int64)
L.oard_debug_timeline_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.oard_debug_timeline_read(buf.ctypes.data, 16) == 0
for w in range(W):
    ev = [(int(x) >> 3, int(x) & 7) for x in buf[w, :9]]
    t0 = ev[0][0]
    print("wave %2d: " % w + "  ".join("%d:%6d" % (c, tt - t0) for tt, c in ev if tt))
