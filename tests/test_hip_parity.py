"""Parity of the HIP path (through the C ABI, via oareactdiff_amd.EGNNDynamics) against
(1) the committed golden outputs of the reference evaluated in float64, and
(2) the CPU oracle, stage by stage.
Tolerance (BASELINE.json north_star): max|ours - ref| / max|ref| <= 1e-5, float32 network."""
import pytest
import torch

import leftnet_oracle as oracle
from _cases import THROUGHPUT, ALL_CASES, LIB_AUTO, Case, debug_options, rel

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


@pytest.mark.parametrize("shapes", ["throughput", "auto"])
@pytest.mark.parametrize("name", ALL_CASES)
def test_forward_matches_reference_f64(name, shapes):
    """shapes = "throughput": the kernels bench.py measures, whatever the size of the case; "auto": the library's
    default launch-shape heuristics, which send these small cases to the latency kernels (oard_edge_small.h)."""
    dev = torch.device("cuda:0")
    c = Case(name)
    dyn = _dyn(c, dev)
    with debug_options(**(LIB_AUTO if shapes == "auto" else {})), torch.no_grad():
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
            ew_got, ew_want = dyn.debug_tap(_capi.TAP_EDGE).cpu(), st[f"l{l}.edgeweight"]
            if l == nl - 1:
                # the last layer's residual update is only evaluated on same-object edges: nothing ever
                # reads the updated state of an inter-object edge (EquiMessage is zero there, leftnet.py:247-249)
                inner = c.n_frag_switch[c.edge_index[0]] == c.n_frag_switch[c.edge_index[1]]
                ew_got, ew_want = ew_got[inner], ew_want[inner]
            assert rel(ew_got, ew_want) <= TOL
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


def test_non_complete_topology_runs_the_general_path():
    """Rounds 1-5 refused every edge list but the complete graph per sample; since round 6 such a call runs the general-edge-list path
    (csrc/oard_general.h; tests/test_general_edges.py) and matches the float64 oracle with the reference's literal node frame."""
    dev = torch.device("cuda:0")
    c = Case("g1_wrapper_small")
    dyn = _dyn(c, dev)
    a = list(_args(c, dev))
    a[1] = a[1][:, :-2]                    # drop two edges
    with torch.no_grad():
        out, _ = dyn(*a)
    assert dyn._last_topo.graph is not None
    ref = oracle.dynamics_forward(c.state_dict(torch.float64), c.cfg, [x.double() for x in c.xh], c.edge_index[:, :-2], c.t.double(),
                                  c.conditions.double(), c.n_frag_switch, c.combined_mask, c.cnf, nodeframe="literal")
    v, h = c.split([o.cpu() for o in out])
    rv, rh = c.split(ref)
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


@pytest.mark.parametrize("name", ["g1_wrapper_small", "g3_cutoff_ragged"])
def test_any_ordering_of_the_complete_edge_set_is_accepted(name):
    """`EGNNDynamics.forward` takes any `edge_index` (egnn_dynamics.py:63-72); outputs are per node, so a PERMUTATION of the edge
    list `get_edges_index` builds (utils/_graph_tools.py:30-36) is the same computation: bit-identical outputs.  A duplicated, a
    missing, a self or a cross-sample edge is a DIFFERENT graph: it runs the general path (round 6) and matches the float64 oracle on
    that edge list, as the reference would; a node id out of range is refused."""
    from oareactdiff_amd._capi import OardError
    dev = torch.device("cuda:0")
    c = Case(name)
    dyn = _dyn(c, dev)
    a = list(_args(c, dev))
    E = a[1].shape[1]
    with torch.no_grad():
        want, _ = dyn(*a)
        for seed in (0, 1):
            perm = torch.randperm(E, generator=torch.Generator().manual_seed(seed)).to(dev)
            b = list(a)
            b[1] = a[1][:, perm].contiguous()
            got, _ = dyn(*b)
            assert all(torch.equal(x, y) for x, y in zip(got, want))
        b[1] = a[1].flip(1).contiguous()                    # reversed order
        got, _ = dyn(*b)
        assert all(torch.equal(x, y) for x, y in zip(got, want))
    bad = []
    dup = a[1].clone(); dup[:, 1] = dup[:, 0]; bad.append(dup)                      # one edge twice, one missing (same count)
    slf = a[1].clone(); slf[1, 0] = slf[0, 0]; bad.append(slf)                      # a self edge
    cm = c.combined_mask
    other = int((cm != cm[int(a[1][0, 0])]).nonzero()[0]) if len(torch.unique(cm)) > 1 else None
    if other is not None:
        crs = a[1].clone(); crs[1, 0] = other; bad.append(crs)                      # an edge between two samples
    for ei in bad:
        b = list(a)
        b[1] = ei
        with torch.no_grad():
            got, _ = dyn(*b)
        assert dyn._last_topo.graph is not None
        ref = oracle.dynamics_forward(c.state_dict(torch.float64), c.cfg, [x.double() for x in c.xh], ei.cpu(), c.t.double(),
                                      c.conditions.double(), c.n_frag_switch, c.combined_mask, c.cnf, nodeframe="literal")
        v, h = c.split([o.cpu() for o in got])
        rv, rh = c.split(ref)
        assert rel(v, rv) <= TOL and rel(h, rh) <= TOL
    oob = a[1].clone(); oob[1, 0] = cm.numel()                                     # node id out of range
    b = list(a)
    b[1] = oob
    with pytest.raises(OardError):
        with torch.no_grad():
            dyn(*b)


def _random_case(sizes, pos_scale, seed, cfg):
    """Ragged synthetic batch (reactions of different atom counts), production feature layout."""
    from oareactdiff_amd.graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch
    g = torch.Generator().manual_seed(seed)
    natm = [torch.tensor(sizes) for _ in range(3)]
    masks = [get_mask_for_frag(n) for n in natm]
    cm = torch.cat(masks)
    nfs = get_n_frag_switch(natm)
    ei = get_edges_index(cm, remove_self_edge=True)
    xh = []
    for m in masks:
        n = m.numel()
        pos = torch.randn(n, 3, generator=g)
        mean = torch.zeros(len(sizes), 3).index_add_(0, m, pos) / torch.tensor(sizes, dtype=torch.float32).unsqueeze(1)
        pos = (pos - mean[m]) * pos_scale
        typ = torch.randint(0, 4, (n,), generator=g)
        feat = torch.zeros(n, 6)
        feat[torch.arange(n), typ] = 1.0
        feat[:, 5] = torch.tensor([1.0, 6.0, 7.0, 8.0])[typ]
        xh.append(torch.cat([pos, feat], 1))
    B = len(sizes)
    return xh, ei, torch.rand(B, 1, generator=g), torch.rand(B, 1, generator=g), nfs, cm


@pytest.mark.parametrize("sizes,pos_scale", [([7, 23, 12, 1, 16], 1.0), ([9, 30, 5], 2.5), ([40, 3], 1.5),
                                             ([1, 1], 1.0), ([2], 1.0)])
@pytest.mark.parametrize("shapes", ["throughput", "auto"])
def test_ragged_production_dims_vs_oracle(sizes, pos_scale, shapes):
    """Ragged reactions (incl. a single-atom-per-object sample and groups > 32 atoms), with and without the
    cutoff biting, production dims, against the float64 oracle evaluated here on the same inputs; the last two
    cases have no same-object edge at all (A = 0) / a single reaction of two-atom objects."""
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=3)
    spec = state_spec(cfg, [9, 9, 9], 1)
    sd = synthetic_state_dict(spec, cfg, seed=7)
    xh, ei, t, cond, nfs, cm = _random_case(sizes, pos_scale, 11, cfg)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    with debug_options(**(LIB_AUTO if shapes == "auto" else {})), torch.no_grad():
        out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    ref = oracle.dynamics_forward({k: v.double() for k, v in sd.items()}, cfg, [x.double() for x in xh], ei,
                                  t.double(), cond.double(), nfs, cm, 1, nodeframe="exact")
    v = torch.cat([o[:, :3].cpu().double().reshape(-1) for o in out])
    h = torch.cat([o[:, 3:].cpu().double().reshape(-1) for o in out])
    rv = torch.cat([o[:, :3].reshape(-1) for o in ref])
    rh = torch.cat([o[:, 3:].reshape(-1) for o in ref])
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL, (rel(v, rv), rel(h, rh))


@pytest.mark.parametrize("sizes,pos_scale", [([9, 30, 5], 2.5), ([40, 3, 17], 4.0), ([23] * 6, 1.0), ([23, 11], 60.0)])
@pytest.mark.parametrize("parts", [1, 2])
def test_skipping_the_inner_edges_outside_the_cutoff_changes_no_bit(sizes, pos_scale, parts):
    """EquiMessage is exactly zero on same-object edges beyond the cutoff (model/leftnet.py:748-753, 768-771: rbf * mask = 0 feeds
    rbf_proj, a Linear without bias).  The edge kernel therefore runs the compacted list of the edges inside the cutoff
    (k_active_list, per call) and the node stage walks the same list, in row order: the sums see the same terms in the same order
    minus exact zeros.  Bit-identical to running every inner row (debug option equi_skip = 0) - with a ragged active set, with
    nothing masked, and with EVERYTHING masked (positions x 60: no active edge at all).  (Throughput launch shapes, pinned: the
    row-lane gathers of small launches deal a node's edges to lanes by their position in the list, so there the two runs agree
    to rounding, not bit for bit.)"""
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=3)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=7)
    xh, ei, t, cond, nfs, cm = _random_case(sizes, pos_scale, 5, cfg)
    outs, n_act = [], []
    for skip in (1, 0):
        with debug_options(equi_skip=skip, parts=parts, **THROUGHPUT), torch.no_grad():
            dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                               condition_nf=1, device=dev)
            dyn.load_state_dict(sd, strict=True)
            out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
            torch.cuda.synchronize()
            n_act.append(dyn.active_inner_edges())
            outs.append([o.clone() for o in out])
    for a, b in zip(*outs):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    A = sum(3 * n * (n - 1) for n in sizes)
    # the count the library reports is the reference's own: distance < cutoff on same-object pairs
    want = 0
    for k in range(3):
        pos = xh[k][:, :3].double()
        m = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
        d = (pos[:, None] - pos[None]).norm(dim=-1)
        want += int(((d < cfg["cutoff"]) & (m[:, None] == m[None])).sum()) - pos.shape[0]
    assert n_act[1] == -1 and n_act[0] == want and 0 <= want <= A, (n_act, want, A)
    if pos_scale >= 60:
        assert want == 0
    if pos_scale == 1.0:
        assert want == A


@pytest.mark.parametrize("pos_scale", [1.0, 3.0])
def test_config5_large_reactions(pos_scale):
    """BASELINE.json configs[4] (SURVEY.md section 8d "config 5"): 128 atoms per object = 384-node reactions with
    147,072 edges each, B=4; positions ~N(0,1) (everything inside the cutoff) and x3 (the cutoff bites: ragged
    active set, one-hop labels).  The first reaction alone is checked against the float64 oracle (3 layers, to
    bound the oracle's memory and time), and the B=4 launch must reproduce it for that reaction."""
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=3)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=7)
    nf, B = 128, 4
    xh, ei, t, cond, nfs, cm = _random_case([nf] * B, pos_scale, 23, cfg)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    with torch.no_grad():
        out4, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    assert all(bool(torch.isfinite(o).all()) for o in out4)
    # the first reaction on its own
    xh1 = [x[:nf].clone() for x in xh]
    natm = [torch.tensor([nf]) for _ in range(3)]
    cm1 = torch.cat([get_mask_for_frag(n) for n in natm])
    nfs1 = get_n_frag_switch(natm)
    ei1 = get_edges_index(cm1, remove_self_edge=True)
    with torch.no_grad():
        out1, _ = dyn([x.to(dev) for x in xh1], ei1.to(dev), t[:1].to(dev), cond[:1].to(dev), nfs1.to(dev), cm1.to(dev))
    # (under the default launch heuristics the two launches use different node-kernel shapes - <= 4 nodes per workgroup walk the
    #  rows of a gather with the wave's columns - i.e. different float32 summation orders: a few 1e-7, not bit-identical)
    for a, b in zip(out1, out4):
        assert rel(b[:nf, :3].cpu(), a[:, :3].cpu()) <= 3e-6 and rel(b[:nf, 3:].cpu(), a[:, 3:].cpu()) <= 3e-6
    ref = oracle.dynamics_forward({k: v.double() for k, v in sd.items()}, cfg, [x.double() for x in xh1], ei1,
                                  t[:1].double(), cond[:1].double(), nfs1, cm1, 1, nodeframe="exact")
    v = torch.cat([o[:, :3].cpu().double().reshape(-1) for o in out1])
    h = torch.cat([o[:, 3:].cpu().double().reshape(-1) for o in out1])
    rv = torch.cat([o[:, :3].reshape(-1) for o in ref])
    rh = torch.cat([o[:, 3:].reshape(-1) for o in ref])
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL, (rel(v, rv), rel(h, rh))


@pytest.mark.parametrize("parts", [1, 0])
def test_large_batch_needs_64bit_offsets(parts):
    """B = 800 reactions on one GPU: 3.75 M edge rows x 688 floats = 2.58e9 elements (10.3 GB), beyond what a
    32-bit element offset can address - as ONE launch (parts=1) and under the default sub-batch schedule.
    Reactions are independent, so the first 64 must reproduce a B=64 launch of the same inputs."""
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    from oareactdiff_amd.synthetic import make_inputs, make_topology
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=7)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    nf, B, b = 23, 800, 64
    g = torch.Generator().manual_seed(3)
    with debug_options(parts=parts):
        cm, nfs, ei, masks = make_topology(B, nf)
        assert ei.shape[1] * 688 > 2 ** 31
        xh = make_inputs(B, nf, masks, 5, dev)
        t, cond = torch.rand(B, 1, generator=g).to(dev), torch.rand(B, 1, generator=g).to(dev)
        with torch.no_grad():
            big, _ = dyn(xh, ei.to(dev), t, cond, nfs.to(dev), cm.to(dev))
        cm2, nfs2, ei2, _ = make_topology(b, nf)
        with torch.no_grad():
            small, _ = dyn([x[:b * nf].clone() for x in xh], ei2.to(dev), t[:b], cond[:b], nfs2.to(dev), cm2.to(dev))
    for x, y in zip(big, small):
        assert bool(torch.isfinite(x).all())
        assert rel(x[:b * nf, :3].cpu(), y[:, :3].cpu()) <= 1e-6 and rel(x[:b * nf, 3:].cpu(), y[:, 3:].cpu()) <= 1e-6
    # and the tail of the batch is computed too (not left as the NaN poison / zeros)
    assert float(big[1][-nf:, :3].abs().max()) > 0


def test_largest_supported_object_and_the_limit():
    """One reaction of 3 x 1024-atom objects (the per-object limit OARD_MAX_GROUP; 9.4 M edges, 26 GB of edge
    state): no oracle runs at that size, so the check is the reference's own property test - a global rotation of
    the input rotates the velocities and leaves the features unchanged - plus finiteness.  One atom more per
    object is served by the general-edge-list path (round 6; rounds 1-5 refused it)."""
    from oareactdiff_amd._capi import OardError
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=1)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=7)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    xh, ei, t, cond, nfs, cm = _random_case([1024], 4.0, 5, cfg)          # x4: the 10 A cutoff bites
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(1)))
    q = q * torch.sign(torch.linalg.det(q))
    xr = [torch.cat([x[:, :3] @ q.T, x[:, 3:]], dim=1) for x in xh]
    args = (ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    with torch.no_grad():
        a, _ = dyn([x.to(dev) for x in xh], *args)
        b, _ = dyn([x.to(dev) for x in xr], *args)
    for x, y in zip(a, b):
        x, y = x.cpu().double(), y.cpu().double()
        assert bool(torch.isfinite(x).all()) and float(x[:, :3].abs().max()) > 0
        assert rel(y[:, :3], x[:, :3] @ q.double().T) <= 1e-4 and rel(y[:, 3:], x[:, 3:]) <= 1e-4
    del a, b
    # One atom more per object is outside the production kernels' tables.  Rounds 1-5 refused it; since round 6 such a layout falls through
    # to the general-edge-list path (csrc/oard_general.h) - here on a sparse edge list (every node to its 8 successors inside its object, both
    # directions), so that the call is cheap: the same rotation property, on the other path.
    xh, _, t, cond, nfs, cm = _random_case([1025], 1.0, 5, cfg)
    src, dst = [], []
    for k in range(3):
        base = torch.arange(1025) + 1025 * k
        for d in range(1, 9):
            src += [base, (base + d - 1025 * k) % 1025 + 1025 * k]
            dst += [(base + d - 1025 * k) % 1025 + 1025 * k, base]
    ei = torch.stack([torch.cat(src), torch.cat(dst)])
    xr = [torch.cat([x[:, :3] @ q.T, x[:, 3:]], dim=1) for x in xh]
    args = (ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    with torch.no_grad():
        a, _ = dyn([x.to(dev) for x in xh], *args)
        assert dyn._last_topo.graph is not None and dyn._last_topo.handle is None
        b, _ = dyn([x.to(dev) for x in xr], *args)
    for x, y in zip(a, b):
        x, y = x.cpu().double(), y.cpu().double()
        assert bool(torch.isfinite(x).all()) and float(x[:, :3].abs().max()) > 0
        assert rel(y[:, :3], x[:, :3] @ q.double().T) <= 1e-4 and rel(y[:, 3:], x[:, 3:]) <= 1e-4


def test_scalar_t_equals_per_sample_t_and_input_is_not_mutated():
    dev = torch.device("cuda:0")
    c = Case("g3_cutoff_ragged")
    dyn = _dyn(c, dev)
    a = list(_args(c, dev))
    keep = [x.clone() for x in a[0]]
    with torch.no_grad():
        o1, _ = dyn(a[0], a[1], torch.tensor([0.25], device=dev), a[3], a[4], a[5])
        B = int(c.combined_mask.max()) + 1
        o2, _ = dyn(a[0], a[1], torch.full((B, 1), 0.25, device=dev), a[3], a[4], a[5])
    for x, y in zip(o1, o2):
        assert torch.equal(x, y)
    for x, y in zip(a[0], keep):
        assert torch.equal(x, y)                       # egnn_dynamics.py:92,97 clone the inputs


def test_condition_and_time_change_the_output():
    """oa_reactdiff/tests/dynamics/test_egnn_dynamics.py:181-228 restated."""
    dev = torch.device("cuda:0")
    c = Case("g1_wrapper_small")
    dyn = _dyn(c, dev)
    a = list(_args(c, dev))
    with torch.no_grad():
        o0, _ = dyn(*a)
        o1, _ = dyn(a[0], a[1], torch.tensor([0.9], device=dev), a[3], a[4], a[5])
        o2, _ = dyn(a[0], a[1], a[2], a[3] + 0.5, a[4], a[5])
    k = 1                                              # object 1 has nodes in both samples
    assert (o0[k] - o1[k]).abs().max() > 1e-6 and (o0[k] - o2[k]).abs().max() > 1e-6


@pytest.mark.parametrize("persist", [0, 2])
@pytest.mark.parametrize("parts", [2, 3])
def test_concurrent_sub_batches_are_bitwise_identical(parts, persist):
    """oard_forward may split a batch into independent sub-batches that run on internal streams; reactions
    never interact, so the result must be bit-identical to the single-part run (with the same GCL edge kernel on both sides:
    by default a single-part launch takes the persistent one, which sums S1 in a different order)."""
    from oareactdiff_amd import _capi
    dev = torch.device("cuda:0")
    c = Case("g3_cutoff_ragged")                      # three samples of different size
    L = _capi.lib()
    outs = []
    try:
        L.oard_debug_option(b"gcl_persist", persist)
        for p in (1, parts):
            L.oard_debug_option(b"parts", p)
            dyn = _dyn(c, dev)
            with torch.no_grad():
                o, _ = dyn(*_args(c, dev))
            torch.cuda.synchronize()
            outs.append([x.clone() for x in o])
    finally:
        L.oard_debug_option(b"parts", 0)
        L.oard_debug_option(b"gcl_persist", 1)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    v, h = c.split([o.cpu() for o in outs[1]])
    rv, rh = c.split(c.ref64)
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


@pytest.mark.parametrize("name", ["g3_cutoff_ragged", "g2_prod_b2_n23"])
def test_persistent_gcl_kernel_against_the_tile_kernel_and_over_grid_sizes(name):
    """k_gcl_edge_p (csrc/oard_edge_p.h: persistent workgroups over half-tiles, balanced last round) against k_gcl_edge_v1 (one
    128-edge tile per workgroup).  Same arithmetic; the one difference is where P[src] + Q[tgt] enters the sum of S1 (behind
    the W1c product instead of in front of it), so the two agree to the last bits, not bit for bit - and the persistent kernel must
    be bit-identical to ITSELF whatever the grid: 1 / 3 / 5 workgroups (shares of many rounds that end in full or half rounds),
    one per CU, two tiles per workgroup (-2); the small cases make every share ragged (rows % 64 != 0, padding wave-tiles)."""
    dev = torch.device("cuda:0")
    c = Case(name)
    rv, rh = c.split(c.ref64)
    outs = {}
    for persist, grid in ((0, 0), (2, 0), (2, 1), (2, 3), (2, 5), (2, -2)):      # 2: also when sub-batches run concurrently
        with debug_options(gcl_persist=persist, gcl_grid=grid):
            dyn = _dyn(c, dev)
            with torch.no_grad():
                o, _ = dyn(*_args(c, dev))
            torch.cuda.synchronize()
            outs[(persist, grid)] = [x.clone() for x in o]
        v, h = c.split([x.cpu() for x in outs[(persist, grid)]])
        assert rel(v, rv) <= TOL and rel(h, rh) <= TOL
    for grid in (1, 3, 5, -2):
        for a, b in zip(outs[(2, 0)], outs[(2, grid)]):
            assert torch.equal(a, b)
    for a, b in zip(outs[(0, 0)], outs[(2, 0)]):
        assert (a - b).abs().max() <= 2e-6 * max(float(a.abs().max()), 1e-6)


@pytest.mark.parametrize("opts", [
    dict(gcl_variant=0, equi_variant=0, node_variant=0),       # v0: weights straight from L2, one wave per 16 nodes
    dict(gcl_variant=0, equi_variant=2, node_variant=0),       # v0 node stages around the streamed EquiMessage kernel
    dict(gcl_variant=3, equi_variant=1),                        # 4-wave workgroups (small launches)
    dict(gcl_variant=2, equi_variant=2, gcl_skip=0),            # no first / last layer shortcuts
    dict(gcl_variant=6, equi_variant=2),                        # latency GCL kernel
    dict(gcl_variant=6, equi_variant=4, gcl_skip=0),            # both latency kernels
])
def test_every_kernel_variant_is_parity_green(opts):
    """The A/B variants kept in the library (oard_debug_option) all compute the same thing.  The first-generation kernels (variant
    0) are compiled into experiment builds only (-DOARD_EXPERIMENTS; `OARD_LIB=.../liboard_exp.so pytest ...`): skipped on a product
    library, which refuses them."""
    from oareactdiff_amd import _capi
    if 0 in (opts.get("gcl_variant"), opts.get("equi_variant"), opts.get("node_variant")) and \
            _capi.lib().oard_debug_option(b"experiments", 1) != 0:
        pytest.skip("first-generation kernels: experiment builds only")
    dev = torch.device("cuda:0")
    with debug_options(**opts):
        for name in ("g2s_prod_b1_n5", "g3_cutoff_ragged", "g2_prod_b2_n23"):
            c = Case(name)
            dyn = _dyn(c, dev)
            with torch.no_grad():
                out, _ = dyn(*_args(c, dev))
            v, h = c.split([o.cpu() for o in out])
            rv, rh = c.split(c.ref64)
            assert rel(v, rv) <= TOL and rel(h, rh) <= TOL, (opts, name, rel(v, rv), rel(h, rh))
