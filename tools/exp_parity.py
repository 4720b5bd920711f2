"""Parity of an experiment-build kernel shape: python tools/exp_parity.py <gcl_variant> [equi_variant]  (OARD_LIB selects the library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
from _cases import Case, rel
from oareactdiff_amd import _capi
from oareactdiff_amd.dynamics import EGNNDynamics
L = _capi.lib()
for k, v in dict(auto_small=0, auto_tiny=0, npb=16, poison=1, gcl_variant=int(sys.argv[1]), equi_variant=int(sys.argv[2]) if len(sys.argv) > 2 else 2).items():
    assert L.oard_debug_option(k.encode(), v) == 0
dev = torch.device("cuda:0")
for name in ("g2_prod_b2_n23", "g3p_prod_cutoff", "g2s_prod_b1_n5"):
    c = Case(name)
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=[f"o{k}" for k in range(c.n_obj)], node_nfs=c.node_nfs, edge_nf=0, condition_nf=c.cnf, device=dev)
    dyn.load_state_dict(c.state_dict(), strict=True)
    with torch.no_grad():
        out, _ = dyn([x.to(dev) for x in c.xh], c.edge_index.to(dev), c.t.to(dev), c.conditions.to(dev), c.n_frag_switch.to(dev), c.combined_mask.to(dev))
    v, h = c.split([o.cpu() for o in out]); rv, rh = c.split(c.ref64)
    print(f"variant {sys.argv[1:]} {name}: vel {rel(v, rv):.2e} h {rel(h, rh):.2e}", "OK" if max(rel(v, rv), rel(h, rh)) <= 1e-5 else "FAIL")
if hasattr(L, "oard_debug_fp_timeouts"):
    import ctypes
    n = ctypes.c_uint(0)
    L.oard_debug_fp_timeouts(ctypes.byref(n))
    print("flag-pipeline wait timeouts:", n.value)
