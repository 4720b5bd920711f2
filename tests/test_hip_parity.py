"""Parity of the HIP path (through the C ABI, via oareactdiff_amd.EGNNDynamics) against
(1) the committed golden outputs of the reference evaluated in float64, and
(2) the CPU oracle, stage by stage.
Tolerance (BASELINE.json north_star): max|ours - ref| / max|ref| <= 1e-5, float32 network."""
import pytest
import torch

import leftnet_oracle as oracle
from _cases import ALL_CASES, Case, rel

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dyn(c, dev):
    from oareactdiff_amd.dynamics import EGNNDynamics
    d = EGNNDynamics(model_config=dict(c.cfg), fragment_names=[f"o{k}" for k in range(c.n_obj)],
                     node_nfs=c.node_nfs, edge_nf=0, condition_nf=c.cnf, device=dev)
    d.load_state_dict(c.state_dict(), strict=True)
    return d


def _args(c, dev):
    return ([x.to(dev) for x in c.xh], c.edge_index.to(dev), c.t.to(dev), c.conditions.to(dev),
            c.n_frag_switch.to(dev), c.combined_mask.to(dev))


@pytest.mark.parametrize("name", ALL_CASES)
def test_forward_matches_reference_f64(name):
    dev = torch.device("cuda:0")
    c = Case(name)
    dyn = _dyn(c, dev)
    with torch.no_grad():
        out, edge_attr = dyn(*_args(c, dev))
    assert edge_attr is None and len(out) == c.n_obj
    for k in range(c.n_obj):
        assert out[k].shape == c.xh[k].shape
    v, h = c.split([o.cpu() for o in out])
    rv, rh = c.split(c.ref64)
    assert rel(v, rv) <= TOL, f"vel {rel(v, rv):.3e}"
    assert rel(h, rh) <= TOL, f"h {rel(h, rh):.3e}"
    # reported beside it, not gated: distance to the reference's own float32 evaluation
    v32, h32 = c.split(c.ref32)
    print(f"{name}: ours-vs-ref64 vel {rel(v, rv):.2e} h {rel(h, rh):.2e} | ours-vs-ref32 vel {rel(v, v32):.2e} "
          f"h {rel(h, h32):.2e} | ref32-vs-ref64 vel {rel(v32, rv):.2e} h {rel(h32, rh):.2e}")


@pytest.mark.parametrize("name", ["g1_wrapper_small", "g2s_prod_b1_n5", "g3_cutoff_ragged", "g3p_prod_cutoff"])
def test_stages_match_oracle(name):
    """Intermediate tensors (taps) against the float64 oracle evaluated on the same inputs."""
    from oareactdiff_amd import _capi
    dev = torch.device("cuda:0")
    c = Case(name)
    dyn = _dyn(c, dev)
    st = {}
    oracle.dynamics_forward(c.state_dict(torch.float64), c.cfg, [x.double() for x in c.xh], c.edge_index,
                            c.t.double(), c.conditions.double(), c.n_frag_switch, c.combined_mask, c.cnf,
                            nodeframe="exact", stages=st)
    H, nl = c.cfg["hidden_channels"], c.cfg["num_layers"]
    L = _capi.lib()
    try:
        L.oard_debug_stop_after(1)
        with torch.no_grad():
            dyn(*_args(c, dev))
        assert rel(dyn.debug_tap(_capi.TAP_POS_FRAME).cpu(), st["pos_frame"]) <= TOL
        lab = dyn.debug_tap(_capi.TAP_LABELS).cpu().long().flatten()
        assert bool(((lab[:, None] == lab[None, :]) == (st["labels"][:, None] == st["labels"][None, :])).all())
        assert rel(dyn.debug_tap(_capi.TAP_S).cpu(), st["s0"]) <= TOL
        assert rel(dyn.debug_tap(_capi.TAP_NE1).cpu(), st["NE1"].reshape(-1, 3 * H)) <= TOL
        assert rel(dyn.debug_tap(_capi.TAP_EDGE).cpu(), st["edgeweight0"]) <= TOL
        for l in (0, nl - 1):
            L.oard_debug_stop_after(100 + 10 * l + 1)
            with torch.no_grad():
                dyn(*_args(c, dev))
            assert rel(dyn.debug_tap(_capi.TAP_S).cpu(), st[f"l{l}.s_gcl"]) <= TOL
            assert rel(dyn.debug_tap(_capi.TAP_EDGE).cpu(), st[f"l{l}.edgeweight"]) <= TOL
            L.oard_debug_stop_after(100 + 10 * l + 2)
            with torch.no_grad():
                dyn(*_args(c, dev))
            assert rel(dyn.debug_tap(_capi.TAP_S).cpu(), st[f"l{l}.s"]) <= TOL
            assert rel(dyn.debug_tap(_capi.TAP_VEC).cpu(), st[f"l{l}.vec"].reshape(-1, 3 * H)) <= TOL
    finally:
        L.oard_debug_stop_after(0)
    with torch.no_grad():
        dyn(*_args(c, dev))
    assert rel(dyn.debug_tap(_capi.TAP_DPOS).cpu(), st["dpos"]) <= TOL
    assert rel(dyn.debug_tap(_capi.TAP_HOUT).cpu(), st["h_out"]) <= TOL


def test_rejects_non_complete_topology():
    from oareactdiff_amd._capi import OardError
    dev = torch.device("cuda:0")
    c = Case("g1_wrapper_small")
    dyn = _dyn(c, dev)
    a = list(_args(c, dev))
    a[1] = a[1][:, :-2]                    # drop two edges
    with pytest.raises(OardError):
        with torch.no_grad():
            dyn(*a)
