"""Which tape entries hold non-finite values after a training-mode forward with the workspace and the tape poisoned (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from oareactdiff_amd import _capi, training
from oareactdiff_amd.dynamics import EGNNDynamics
from _grad_cases import CNF, NODE_NFS, GradCase

L = _capi.lib()
L.oard_debug_option(b"poison", 1)
name = sys.argv[1] if len(sys.argv) > 1 else "g9_grad_prod_n23"
c = GradCase(name)
dev = torch.device("cuda:0")
dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=NODE_NFS, edge_nf=0, condition_nf=CNF, device=dev)
dyn.load_state_dict(c.state_dict(), strict=True)
keep = {}
orig = training.DynamicsFunction.forward
def spy(ctx, dyn_, run_forward, n_obj, *tensors):
    o = orig(ctx, dyn_, run_forward, n_obj, *tensors)
    keep["state"] = ctx.state
    return o
training.DynamicsFunction.forward = staticmethod(spy)
loss = c.loss(dyn, torch.float32, dev)
training.DynamicsFunction.forward = orig
st = keep["state"]; tape, topo = st.tape, st.topo
N, E, A = topo.N, topo.E, topo.A
print("loss", float(loss), "N E A", N, E, A)
NL = c.cfg["num_layers"]
names = {v: k for k, v in vars(_capi).items() if k.startswith("TAPE_")}
for which in range(16, 29):
    for l in range(NL + 1):
        try:
            t = tape.get(which, l)
        except Exception:
            continue
        rows = {E + 1: E, A + 1: A}.get(t.shape[0], t.shape[0])
        bad = (~torch.isfinite(t[:rows])).any(dim=1)
        if bool(bad.any()):
            idx = torch.nonzero(bad).flatten()
            print(names[which], "layer", l, "shape", tuple(t.shape), "bad rows", int(bad.sum()), "first", idx[:6].tolist(), "last", idx[-3:].tolist())
