"""General edge lists on the GPU (csrc/oard_general.h / oard_general.hip; include/oard.h "general edge lists"): EGNNDynamics.forward on
edge lists that are not the complete graph per sample - what the reference accepts (dynamics/egnn_dynamics.py:63-72), builds with
`edge_cutoff` (utils/_graph_tools.py:31-33) and exercises in its own model tests (tests/model/test_equiv.py:177-230,
tests/model/test_subgraphs.py:285-339).  Gate: 1e-5 of the reference evaluated in float64 (fixtures g11_* made from the reference by
oracle/make_goldens.py; larger graphs against the oracle on the box, which the fixtures pin to 3e-16 on general graphs)."""
import pytest
import torch

import leftnet_oracle as oracle
from _cases import ALL_CASES, Case, rel

pytestmark = pytest.mark.gpu
TOL = 1e-5
GENERAL_CASES = ["g11_edge_cutoff_h32", "g11p_edge_cutoff_prod", "g11_random_subset", "g11_components_noreflect", "g11_arbitrary"]


def _dyn(c, dev, path="auto"):
    from oareactdiff_amd.dynamics import EGNNDynamics
    d = EGNNDynamics(model_config=dict(c.cfg), fragment_names=[f"o{k}" for k in range(c.n_obj)], node_nfs=c.node_nfs, edge_nf=0,
                     condition_nf=c.cnf, device=dev)
    d.load_state_dict(c.state_dict(), strict=True)
    d.edge_list_path = path
    return d


def _args(c, dev, ei=None):
    return ([x.to(dev) for x in c.xh], (c.edge_index if ei is None else ei).to(dev), c.t.to(dev), c.conditions.to(dev),
            c.n_frag_switch.to(dev), c.combined_mask.to(dev))


@pytest.mark.parametrize("name", GENERAL_CASES)
def test_general_edge_lists_match_the_reference_f64(name):
    dev = torch.device("cuda:0")
    c = Case(name)
    dyn = _dyn(c, dev)
    with torch.no_grad():
        out, _ = dyn(*_args(c, dev))
    assert dyn._last_topo.graph is not None and dyn._last_topo.handle is None      # it WAS the general path
    v, h = c.split([o.cpu() for o in out])
    rv, rh = c.split(c.ref64)
    r32v, r32h = c.split(c.ref32)
    print(f"{name}: E {c.edge_index.shape[1]}  vel {rel(v, rv):.2e} h {rel(h, rh):.2e}   (the reference's own float32: vel {rel(r32v, rv):.1e} "
          f"h {rel(r32h, rh):.1e})")
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


@pytest.mark.parametrize("name", ALL_CASES)
def test_general_path_agrees_with_the_production_kernels_on_complete_graphs(name):
    """Two independent implementations of the same network: the production kernels (implicit complete graph, exact-arithmetic node frame,
    MFMA chains) and the general path (explicit edge list, literal node frame on float64 geometry) on the eight complete-graph fixtures."""
    dev = torch.device("cuda:0")
    c = Case(name)
    with torch.no_grad():
        a, _ = _dyn(c, dev)(*_args(c, dev))
        g = _dyn(c, dev, "general")
        b, _ = g(*_args(c, dev))
    assert g._last_topo.graph is not None
    va, ha = c.split([o.cpu() for o in a])
    vb, hb = c.split([o.cpu() for o in b])
    rv, rh = c.split(c.ref64)
    print(f"{name}: general vs f64 reference vel {rel(vb, rv):.2e} h {rel(hb, rh):.2e}; general vs production vel {rel(vb, va):.2e} h {rel(hb, ha):.2e}")
    assert rel(vb, rv) <= TOL and rel(hb, rh) <= TOL
    assert rel(vb, va) <= TOL and rel(hb, ha) <= TOL


@pytest.mark.parametrize("name", ["g11p_edge_cutoff_prod", "g11_arbitrary", "g3p_prod_cutoff"])
def test_dense_layers_on_the_matrix_pipe_and_on_plain_threads_agree(name, monkeypatch):
    """The dense layers of the general path run on the float64 matrix pipe (k_general_gemm_f64: 16-row x 16-output MFMA tiles, ragged row and
    output tails, gathered segments); OARD_GENERAL_GEMM=threads runs the same layers as og::Gemm on plain threads - the formulation the CPU
    suite checks on the host executor.  Both carry the sum in float64: they may differ in its ORDER only.  (The kernel reads float4s: the layers
    whose segments are not multiples of four columns - pos_expansion's H / 2 = 98 and 3 inputs, the embeddings' in_hidden + 1 = 9 - stay on
    plain threads in either setting, so every fixture at production width runs a mix of the two.)"""
    dev = torch.device("cuda:0")
    c = Case(name)
    outs = {}
    for how in ("matrix", "threads"):
        monkeypatch.setenv("OARD_GENERAL_GEMM", how)                       # read by the library on every call
        dyn = _dyn(c, dev, "general")
        with torch.no_grad():
            out, _ = dyn(*_args(c, dev))
        assert dyn._last_topo.graph is not None
        outs[how] = c.split([o.cpu() for o in out])
    (va, ha), (vb, hb) = outs["matrix"], outs["threads"]
    rv, rh = c.split(c.ref64)
    print(f"{name}: matrix pipe vs threads vel {rel(va, vb):.1e} h {rel(ha, hb):.1e}; vs f64 reference: matrix vel {rel(va, rv):.2e}, threads vel {rel(vb, rv):.2e}")
    assert rel(va, vb) <= 2e-7 and rel(ha, hb) <= 2e-7
    assert rel(va, rv) <= TOL and rel(vb, rv) <= TOL


@pytest.mark.parametrize("hidden, radial", [(36, 6), (40, 12), (64, 20)])
def test_widths_nobody_built_run_the_general_kernels(hidden, radial):
    """The production kernels exist per (hidden_channels, num_radial) pair of the build (OARD_DIMS_LIST); the general path's widths are
    run-time values.  A module of any other width - the reference's tests and notebooks use many - answers inference calls through the general
    kernels (a warning, once, names the rebuild), on complete and cut graphs alike; training at such a width still raises.  Against the float64
    oracle evaluated here.  (36, 6): H / 2 = 18, num_radial = 6 and an edge width of 114 are not multiples of four - most dense layers of that
    network stay on plain threads, a few run on the matrix pipe; (40, 12) and (64, 20): nearly all on the matrix pipe.)"""
    from oareactdiff_amd import _capi
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2, hidden_channels=hidden, num_radial=radial)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=3)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    natm = [torch.tensor([5, 9, 3]) for _ in range(3)]
    masks = [get_mask_for_frag(n) for n in natm]
    cm, nfs = torch.cat(masks), get_n_frag_switch(natm)
    g = torch.Generator().manual_seed(hidden + radial)
    xh = [torch.cat([2.0 * torch.randn(m.numel(), 3, generator=g), torch.rand(m.numel(), 6, generator=g)], 1) for m in masks]
    full = get_edges_index(cm, remove_self_edge=True)
    cut = get_edges_index(cm, pos=torch.cat([x[:, :3] for x in xh]), edge_cutoff=4.0, remove_self_edge=True)
    assert 0 < cut.shape[1] < full.shape[1]
    t, cond = torch.rand(3, 1, generator=g), torch.rand(3, 1, generator=g)
    for name, ei in (("complete", full), ("edge_cutoff", cut)):
        with torch.no_grad():
            if name == "complete":
                with pytest.warns(UserWarning, match="not a width pair the production kernels were built for"):
                    out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
            else:
                out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
        assert dyn._last_topo.graph is not None and dyn._last_topo.handle is None
        ref = oracle.dynamics_forward({k: v.double() for k, v in sd.items()}, cfg, [x.double() for x in xh], ei, t.double(), cond.double(), nfs, cm, 1,
                                      nodeframe="literal")
        v = torch.cat([o[:, :3].cpu().double().reshape(-1) for o in out]); h = torch.cat([o[:, 3:].cpu().double().reshape(-1) for o in out])
        rv = torch.cat([o[:, :3].reshape(-1) for o in ref]); rh = torch.cat([o[:, 3:].reshape(-1) for o in ref])
        print(f"H {hidden} R {radial}, {name}: {ei.shape[1]} edges, vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
        assert rel(v, rv) <= TOL and rel(h, rh) <= TOL
    with pytest.raises(_capi.OardError, match="not built"):                 # under autograd: the backward pass exists for built widths only
        dyn([x.to(dev) for x in xh], full.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))


def test_edge_cutoff_batch_against_the_oracle():
    """get_edges_index(combined_mask, pos, edge_cutoff) (utils/_graph_tools.py:31-33) on a ragged batch at production dims, against the
    float64 oracle (literal node frame) evaluated here."""
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=3)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=11)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    natm = [torch.tensor([7, 12, 4, 9]) for _ in range(3)]
    masks = [get_mask_for_frag(n) for n in natm]
    cm, nfs = torch.cat(masks), get_n_frag_switch(natm)
    g = torch.Generator().manual_seed(5)
    xh = [torch.cat([2.0 * torch.randn(m.numel(), 3, generator=g), torch.rand(m.numel(), 6, generator=g)], 1) for m in masks]
    pos = torch.cat([x[:, :3] for x in xh])
    ei = get_edges_index(cm, pos=pos, edge_cutoff=4.5, remove_self_edge=True)
    full = get_edges_index(cm, remove_self_edge=True).shape[1]
    assert 0 < ei.shape[1] < full
    t, cond = torch.rand(4, 1, generator=g), torch.rand(4, 1, generator=g)
    with torch.no_grad():
        out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    ref = oracle.dynamics_forward({k: v.double() for k, v in sd.items()}, cfg, [x.double() for x in xh], ei, t.double(), cond.double(), nfs, cm, 1,
                                  nodeframe="literal")
    v = torch.cat([o[:, :3].cpu().double().reshape(-1) for o in out]); h = torch.cat([o[:, 3:].cpu().double().reshape(-1) for o in out])
    rv = torch.cat([o[:, :3].reshape(-1) for o in ref]); rh = torch.cat([o[:, 3:].reshape(-1) for o in ref])
    print(f"edge_cutoff batch: {ei.shape[1]} of {full} edges, vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


def test_parts_without_edges_between_them_do_not_interact():
    """tests/model/test_equiv.py:216-230 (a disconnected part computes what it computes alone) and tests/model/test_subgraphs.py:285-339 (no
    e_ij between two parts: moving one changes nothing in the other), through the dynamics wrapper: object 2 of every sample keeps its
    internal edges but has no edge to objects 0 and 1.  Replacing objects 0 / 1 by other molecules leaves object 2's outputs unchanged to
    the last bit; the connected objects do change."""
    dev = torch.device("cuda:0")
    c = Case("g3_cutoff_ragged")
    nfs = c.n_frag_switch
    a, b = c.edge_index
    keep = (nfs[a] == 2) == (nfs[b] == 2)              # drop every edge between object 2 and the others
    ei = c.edge_index[:, keep]
    assert 0 < ei.shape[1] < c.edge_index.shape[1]
    dyn = _dyn(c, dev)
    args = list(_args(c, dev, ei))
    with torch.no_grad():
        out, _ = dyn(*args)
        moved = [x.clone() for x in args[0]]
        g = torch.Generator().manual_seed(1)
        for k in (0, 1):
            moved[k] = torch.cat([torch.randn(moved[k].shape[0], 3, generator=g), torch.rand(moved[k].shape[0], moved[k].shape[1] - 3,
                                                                                             generator=g)], 1).to(dev)
        args2 = list(args)
        args2[0] = moved
        out2, _ = dyn(*args2)
    assert torch.equal(out[2], out2[2])
    assert not torch.equal(out[0], out2[0]) and not torch.equal(out[1], out2[1])
    # ... and the isolated object is what the float64 oracle computes on this edge list
    ref = oracle.dynamics_forward(c.state_dict(torch.float64), c.cfg, [x.double() for x in c.xh], ei, c.t.double(), c.conditions.double(), nfs,
                                  c.combined_mask, c.cnf, nodeframe="literal")
    assert rel(out[2].cpu(), ref[2]) <= TOL


def test_what_is_still_refused():
    """Node ids out of range; training (the tape and the hand-written backward are built on the complete graph)."""
    from oareactdiff_amd._capi import OardError
    dev = torch.device("cuda:0")
    c = Case("g11_random_subset")
    dyn = _dyn(c, dev)
    bad = c.edge_index.clone()
    bad[1, 0] = c.combined_mask.numel()
    with pytest.raises(OardError), torch.no_grad():
        dyn(*_args(c, dev, bad))
    with pytest.raises(OardError):
        dyn(*_args(c, dev))                                # autograd enabled: the training path


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_arbitrary_edge_lists_are_taken_as_given(seed):
    """Self loops, duplicated edges, edges across samples and objects, isolated nodes, an empty object: whatever the list holds is the
    graph, as in the reference (egnn_dynamics.py:63-72; the oracle, pinned on the reference for general graphs, evaluates the same list)."""
    from test_general_host import _random_graph_case
    dev = torch.device("cuda:0")
    c = _random_graph_case(seed)
    dyn = _dyn(c, dev)
    with torch.no_grad():
        out, _ = dyn(*_args(c, dev))
    assert dyn._last_topo.graph is not None
    ref = oracle.dynamics_forward(c.state_dict(torch.float64), c.cfg, [x.double() for x in c.xh], c.edge_index, c.t.double(),
                                  c.conditions.double(), c.n_frag_switch, c.combined_mask, c.cnf, nodeframe="literal")
    v, h = c.split([o.cpu() for o in out])
    rv, rh = c.split(ref)
    print(f"random graph {seed}: vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL

