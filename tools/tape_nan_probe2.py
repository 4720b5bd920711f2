"""As tape_nan_probe.py on the bench's training batch (B reactions x 23 atoms): non-finite tape rows after one fused forward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_training_batch, new_dynamics
from oareactdiff_amd import _capi
from oareactdiff_amd.trainer import DDPMTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = _capi.lib()
L.oard_debug_option(b"poison", 1)
dev = torch.device("cuda:0")
dyn = new_dynamics(dev)
tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
keep = {}
orig = dyn._run_forward_train
def spy(*a, **k):
    net, state = orig(*a, **k)
    keep["state"] = state
    return net, state
dyn._run_forward_train = spy
torch.manual_seed(1234)
tr._bucket.zero_()
batch = make_training_batch(B, 23, 100, dev)
nll, terms = tr._fused_forward_backward(batch)
torch.cuda.synchronize()
st = keep["state"]; tape, topo = st.tape, st.topo
N, E, A = topo.N, topo.E, topo.A
print("nll finite", bool(torch.isfinite(nll).all()), "grad finite", bool(torch.isfinite(tr.flat_grad).all()), "N E A", N, E, A)
names = {v: k for k, v in vars(_capi).items() if k.startswith("TAPE_")}
for which in range(16, 29):
    for l in range(7):
        try:
            t = tape.get(which, l)
        except Exception:
            continue
        rows = {E + 1: E, A + 1: A}.get(t.shape[0], t.shape[0])
        bad = (~torch.isfinite(t[:rows])).any(dim=1)
        if bool(bad.any()):
            idx = torch.nonzero(bad).flatten()
            print(names[which], "layer", l, "shape", tuple(t.shape), "bad rows", int(bad.sum()), "first", idx[:4].tolist(), "last", idx[-3:].tolist())
