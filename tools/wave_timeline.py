"""Per-wave timeline of one k_gcl_edge_v1 workgroup in an experiment build (-DOARD_EXPERIMENTS -DOARD_TIMELINE):
    OARD_LIB=.../liboard_tl.so python tools/wave_timeline.py [out.npy]
prints, per phase kind, when each wave passes the barrier / starts and ends its chains (cycles relative to the phase start)."""
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
for k, v in dict(parts=1, sequential=1, gcl_variant=int(os.environ.get("OARD_GCL_VARIANT", "2"))).items():
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
W, TL_MAX = 8, 1024
buf = np.zeros((W, TL_MAX), dtype=np.int64)
L.oard_debug_timeline_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.oard_debug_timeline_read(buf.ctypes.data, W) == 0
if len(sys.argv) > 1:
    np.save(sys.argv[1], buf)
ev = []
for w in range(W):
    n = int(np.argmax(buf[w] == 0)) if (buf[w] == 0).any() else TL_MAX
    ev.append([(int(x) >> 3, int(x) & 7) for x in buf[w, :n]])
t0 = min(e[0][0] for e in ev)
print("events per wave:", [len(e) for e in ev], " kernel span (cycles):", max(e[-1][0] for e in ev) - t0)
# split into phases at code 1
phases = [[] for _ in range(W)]
for w in range(W):
    cur = None
    for tt, c in ev[w]:
        if c == 1:
            cur = [tt]
            phases[w].append(cur)
        elif cur is not None:
            cur.append(tt)
for w in range(W):
    print(f"wave {w}: entry -> first barrier exit {ev[w][1][0] - ev[w][0][0]} cycles; last event -> {ev[w][-1][0] - ev[w][0][0]} (code {ev[w][-1][1]})")
nph = min(len(p) for p in phases)
lens = [min(phases[w][ph + 1][0] for w in range(W)) - min(phases[w][ph][0] for w in range(W)) for ph in range(nph - 1)]
print("phase lengths:", lens, "sum", sum(lens))
print("phases:", nph)
for ph in list(range(0, 4)) + list(range(22, 30)) + list(range(nph - 4, nph)):
    base = min(phases[w][ph][0] for w in range(W))
    nxt = min(phases[w][ph + 1][0] for w in range(W)) if ph + 1 < nph else None
    print(f"phase {ph}: length {'' if nxt is None else nxt - base}")
    for w in range(W):
        print("   wave", w, " ".join(f"{x - base:6d}" for x in phases[w][ph]))
