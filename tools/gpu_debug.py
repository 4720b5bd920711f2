"""Stage-by-stage comparison of the HIP path against the CPU oracle (float64, exact node frame)
on the golden cases.  Run on the GPU box:  python tools/gpu_debug.py [case ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import leftnet_oracle as oracle  # noqa: E402
from _cases import ALL_CASES, Case, rel  # noqa: E402
from oareactdiff_amd import _capi  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402


def build(c: Case, dev):
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=[f"o{k}" for k in range(c.n_obj)],
                       node_nfs=c.node_nfs, edge_nf=0, condition_nf=c.cnf, device=dev)
    dyn.load_state_dict(c.state_dict(), strict=True)
    return dyn.to(dev)


def main():
    names = sys.argv[1:] or ALL_CASES
    dev = torch.device("cuda:0")
    L = _capi.lib()
    for name in names:
        c = Case(name)
        print(f"=== {name}  N={c.combined_mask.numel()} E={c.edge_index.shape[1]} cfg={c.cfg}")
        sd64 = c.state_dict(torch.float64)
        st = {}
        o64 = oracle.dynamics_forward(sd64, c.cfg, [x.double() for x in c.xh], c.edge_index, c.t.double(),
                                      c.conditions.double(), c.n_frag_switch, c.combined_mask, c.cnf,
                                      nodeframe="exact", stages=st)
        dyn = build(c, dev)
        dyn.nan_check = "async"
        args = ([x.to(dev) for x in c.xh], c.edge_index.to(dev), c.t.to(dev), c.conditions.to(dev),
                c.n_frag_switch.to(dev), c.combined_mask.to(dev))
        H = c.cfg["hidden_channels"]
        nl = c.cfg["num_layers"]

        def run(stop):
            L.oard_debug_stop_after(stop)
            with torch.no_grad():
                out, _ = dyn(*args)
            torch.cuda.synchronize()
            L.oard_debug_stop_after(0)
            return out

        def show(tag, got, want):
            got = got.detach().cpu().double().reshape(want.shape)
            bad = "" if torch.isfinite(got).all() else "  NON-FINITE"
            print(f"   {tag:24s} rel={rel(got, want):.3e}  max|ref|={float(want.abs().max()):.3e}{bad}")

        run(1)
        show("pos_frame", dyn.debug_tap(_capi.TAP_POS_FRAME), st["pos_frame"])
        lab = dyn.debug_tap(_capi.TAP_LABELS).cpu().long().flatten()
        same_o = st["labels"][:, None] == st["labels"][None, :]
        same_g = lab[:, None] == lab[None, :]
        print(f"   labels partition equal: {bool((same_o == same_g).all())}")
        show("s0", dyn.debug_tap(_capi.TAP_S), st["s0"])
        show("NE1", dyn.debug_tap(_capi.TAP_NE1), st["NE1"].reshape(-1, 3 * H))
        show("edgeweight0", dyn.debug_tap(_capi.TAP_EDGE), st["edgeweight0"])
        ew0 = dyn.debug_tap(_capi.TAP_EDGE).cpu().double()
        w = st["edgeweight0"]
        for nm, sl in (("  ew0[sc3]", slice(0, H)), ("  ew0[sc4]", slice(H, 2 * H)), ("  ew0[f]", slice(2 * H, 3 * H)),
                       ("  ew0[rbf]", slice(3 * H, None))):
            print(f"   {nm:24s} rel={rel(ew0[:, sl], w[:, sl]):.3e}")
        for l in range(nl):
            run(100 + 10 * l + 1)
            show(f"l{l}.s_gcl", dyn.debug_tap(_capi.TAP_S), st[f"l{l}.s_gcl"])
            inner = c.n_frag_switch[c.edge_index[0]] == c.n_frag_switch[c.edge_index[1]]
            show(f"l{l}.edgeweight[inner]", dyn.debug_tap(_capi.TAP_EDGE).cpu()[inner], st[f"l{l}.edgeweight"][inner])
            if l < nl - 1:
                show(f"l{l}.edgeweight[all]", dyn.debug_tap(_capi.TAP_EDGE), st[f"l{l}.edgeweight"])
            run(100 + 10 * l + 2)
            show(f"l{l}.s", dyn.debug_tap(_capi.TAP_S), st[f"l{l}.s"])
            show(f"l{l}.vec", dyn.debug_tap(_capi.TAP_VEC), st[f"l{l}.vec"].reshape(-1, 3 * H))
        out = run(0)
        show("dpos", dyn.debug_tap(_capi.TAP_DPOS), st["dpos"])
        show("h_out", dyn.debug_tap(_capi.TAP_HOUT), st["h_out"])
        v, h = c.split([o.cpu() for o in out])
        rv, rh = c.split(c.ref64)
        print(f"   FINAL vs reference f64: vel rel={rel(v, rv):.3e}  h rel={rel(h, rh):.3e}   status={dyn.last_status.tolist()}")


if __name__ == "__main__":
    main()
