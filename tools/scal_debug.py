import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
from _grad_cases import CNF, NODE_NFS, GradCase
from oareactdiff_amd import _capi, training
from oareactdiff_amd.dynamics import EGNNDynamics
c = GradCase("g9_grad_h32")
dev = torch.device("cuda:0")
dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=NODE_NFS, edge_nf=0, condition_nf=CNF, device=dev)
dyn.load_state_dict(c.state_dict(), strict=True)
keep = {}
orig = training.DynamicsFunction.forward
def spy(ctx, dyn_, run_forward, n_obj, *tensors):
    out = orig(ctx, dyn_, run_forward, n_obj, *tensors); keep["state"] = ctx.state; return out
training.DynamicsFunction.forward = staticmethod(spy)
loss = c.loss(dyn, torch.float32, dev)
st = keep["state"]; tape, topo, cfg = st.tape, st.topo, st.cfg
H, R, NL, Cc = dyn._dims
WP = training._pad16(3 * H + R)
N, E, A = topo.N, topo.E, topo.A
P = dyn._param_dict()
geo = tape.get(_capi.TAPE_GEO)[:A]
g = training.Geometry(topo.inner_src, topo.inner_tgt, topo.node_sample, topo.node_group, topo.B, topo.B * 3, geo,
                      tape.get(_capi.TAPE_RBF)[:A, :R], tape.get(_capi.TAPE_PP0)[:, 0], tape.get(_capi.TAPE_X1))
stream = torch.cuda.current_stream(dev).cuda_stream
with torch.no_grad():
    hin = tape.get(_capi.TAPE_HIN)[:, :Cc]
    _, NE1, _, _ = training.stage_init_head(P, hin, g, H)
print("groups", topo.node_group.tolist()); print("src", topo.inner_src.tolist()); print("tgt", topo.inner_tgt.tolist())
for mode in ("both", "side0", "side1", "residual_only"):
    gen0 = torch.Generator(device="cpu").manual_seed(11)
    Gs = torch.randn(A, 2 * H, generator=gen0).to(dev)
    if mode == "side0": Gs[:, H:] = 0
    if mode == "side1": Gs[:, :H] = 0
    NE1t = NE1.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        sc = training.stage_scalarize(P, NE1t, g, H)
    gs = torch.autograd.grad([sc], [NE1t], [Gs])[0]
    dews = torch.zeros(E + 1, WP, device=dev); dews[:A, :2 * H] = Gs
    dNE1, gl3 = training.scalarize_backward(dyn, cfg, topo, tape, NE1.contiguous(), dews, H, stream)
    err = (dNE1 - gs).abs().amax(dim=(1, 2)) / gs.abs().max()
    print(mode, "max", float(err.max()), "per node", [round(float(x), 4) for x in err])
    print("   per channel", [round(float(x), 4) for x in ((dNE1 - gs).abs().amax(dim=(0, 1)) / gs.abs().max())])
    print("   per xyz", [round(float(x), 4) for x in ((dNE1 - gs).abs().amax(dim=(0, 2)) / gs.abs().max())])
