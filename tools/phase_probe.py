"""Cycle accounting of k_gcl_edge_v1's phase protocol in a probe build (run on the GPU box):
    OARD_LIB=.../liboard_probe.so OARD_CXXFLAGS="-DOARD_EXPERIMENTS -DOARD_PHASE_PROBE" python -m oareactdiff_amd.build
    OARD_LIB=.../liboard_probe.so python tools/phase_probe.py [parts] [sequential] [gcl_variant]
Prints, per wave and phase: cycles in the s_waitcnt before the phase barrier, in the barrier, per LDS-DMA issue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oareactdiff_amd import _capi
from oareactdiff_amd.dynamics import EGNNDynamics
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
from oareactdiff_amd.synthetic import make_inputs, make_topology

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B = int(os.environ.get("PROBE_B", "64"))
L = _capi.lib()
for k, v in dict(parts=parts, sequential=seq, gcl_variant=variant).items():
    assert L.oard_debug_option(k.encode(), v) == 0
L.oard_debug_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
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
buf = (ctypes.c_ulonglong * 8)()
def step():
    with torch.no_grad():
        dyn(inp, ei, t, cond, nfs, cm)
for _ in range(3):
    step()
assert L.oard_debug_probe_read(buf) == 0
n = 5
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    step()
e1.record()
torch.cuda.synchronize()
assert L.oard_debug_probe_read(buf) == 0
waves, total, wait, bar, dcyc, dn, nph, bar_lo = [int(x) for x in buf[:8]]
print(f"B={B} parts={parts} sequential={seq} gcl_variant={variant}: {e0.elapsed_time(e1) / n:.3f} ms/step (probe build)")
print(f"  waves/step {waves / n:.0f}  phases/wave {nph / waves:.1f}  cycles/wave {total / waves:.0f}")
print(f"  per phase: total {total / nph:.0f}  waitcnt {wait / nph:.0f} ({100 * wait / total:.1f} %)  barrier {bar / nph:.0f} ({100 * bar / total:.1f} %)")
print(f"  barrier cycles spent by the lower half of the waves of a workgroup (wave < WAVES/2): {100 * bar_lo / max(bar, 1):.1f} %")
print(f"  LDS-DMA: {dn / waves:.1f} pieces/wave, {dcyc / max(dn, 1):.0f} cycles per issue ({100 * dcyc / total:.1f} % of wave time)")
