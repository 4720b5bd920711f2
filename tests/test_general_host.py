"""The general-edge-list path's formulas on the CPU (csrc/oard_general.h).  The stage functors and the orchestration of that path are
written once for two executors; here tests/general_host/harness.cpp runs them in host loops (compiled with g++ by this test) against the
REFERENCE's float64 outputs (tests/golden/*.npz, oracle/make_goldens.py): the eight complete-graph fixtures - where the literal node
frame this path evaluates agrees with the exact-arithmetic one - and the five general graphs of round 6 (edge_cutoff graphs from the
reference's own builder, a directed subset in arbitrary order, disconnected components with reflect_equiv=False, an arbitrary list with
self loops / duplicates / cross-sample edges).
On the GPU the same functors run as HIP kernels (tests/test_general_edges.py, -m gpu).  Nothing in the product path loads this harness."""
import ctypes as C
import os
import subprocess

import pytest
import torch

from _cases import ALL_CASES, Case, rel

HERE = os.path.dirname(os.path.abspath(__file__))
GENERAL_CASES = ["g11_edge_cutoff_h32", "g11p_edge_cutoff_prod", "g11_random_subset", "g11_components_noreflect", "g11_arbitrary"]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("general_host") / "libgeneral_host.so")
    # OARD_HOST_SANITIZE=1 (with LD_PRELOAD=$(gcc -print-file-name=libasan.so) for the interpreter): the same functors under ASan + UBSan
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if os.environ.get("OARD_HOST_SANITIZE") else ["-O2"]
    subprocess.run(["g++", *san, "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "general_host", "harness.cpp")], check=True)
    return C.CDLL(so)


def run_host(harness, c: Case, edge_index=None):
    from oareactdiff_amd.dynamics import EGNNDynamics
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=[f"o{k}" for k in range(c.n_obj)], node_nfs=c.node_nfs, edge_nf=0,
                       condition_nf=c.cnf, device=torch.device("cpu"))
    dyn.load_state_dict(c.state_dict(), strict=True)
    cfg = dyn._config()
    tensors = [t.detach().float().contiguous() if t is not None else None for t in dyn._ordered_tensors()]
    vp = C.c_void_p
    params = (vp * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])
    xh = [x.float().contiguous() for x in c.xh]
    out = [torch.full_like(x, float("nan")) for x in xh]
    ei = (c.edge_index if edge_index is None else edge_index).long().contiguous()
    cm, nfs = c.combined_mask.long().contiguous(), c.n_frag_switch.long().contiguous()
    t, cond = c.t.float().contiguous(), c.conditions.float().contiguous()
    status = torch.zeros(1, dtype=torch.int32)
    arr = lambda ts: (vp * len(ts))(*[x.data_ptr() for x in ts])                # noqa: E731
    rc = harness.oard_general_forward_host(C.byref(cfg), vp(cm.data_ptr()), vp(nfs.data_ptr()), C.c_int64(cm.numel()), vp(ei.data_ptr()),
                                           C.c_int64(ei.shape[1]), params, C.c_size_t(len(tensors)), arr(xh), vp(t.data_ptr()),
                                           1 if t.dim() == 1 else 0, vp(cond.data_ptr()), arr(out), vp(status.data_ptr()))
    assert rc == 0, rc
    assert int(status[0]) == 0
    return out


@pytest.mark.parametrize("name", ALL_CASES + GENERAL_CASES)
def test_general_path_formulas_match_the_reference_f64(harness, name):
    c = Case(name)
    out = run_host(harness, c)
    v, h = c.split(out)
    rv, rh = c.split(c.ref64)
    print(f"{name}: N {c.combined_mask.numel()} E {c.edge_index.shape[1]}  vel {rel(v, rv):.2e}  h {rel(h, rh):.2e}")
    assert all(bool(torch.isfinite(o).all()) for o in out)
    assert rel(v, rv) <= 1e-5 and rel(h, rh) <= 1e-5


def test_general_path_is_invariant_under_edge_order(harness):
    """Per-node outputs; every aggregation runs in edge order, so a permutation moves the float64 sums' last bits only."""
    c = Case("g11_edge_cutoff_h32")
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(c.edge_index.shape[1], generator=g)
    a, b = run_host(harness, c), run_host(harness, c, c.edge_index[:, perm])
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max())


def _random_graph_case(seed):
    """A small ragged batch with an ARBITRARY directed edge list: self loops, duplicated edges, edges across samples and across objects,
    nodes without incoming / outgoing edges, an empty object - everything `EGNNDynamics.forward` takes as given (egnn_dynamics.py:63-72)."""
    from oareactdiff_amd.graph_tools import get_mask_for_frag, get_n_frag_switch
    g = torch.Generator().manual_seed(seed)
    c = Case("g3_cutoff_ragged")                       # configuration / weights of a fixture; inputs and graph are drawn here
    sizes = [[3, 2, 4], [2, 0, 3], [1, 3, 2]]
    natm = [torch.tensor(x) for x in sizes]
    masks = [get_mask_for_frag(n) for n in natm]
    cm, nfs = torch.cat(masks), get_n_frag_switch(natm)
    N = cm.numel()
    c.xh = [torch.cat([1.5 * torch.randn(m.numel(), 3, generator=g), torch.rand(m.numel(), 6, generator=g)], 1) for m in masks]
    E = 70
    ei = torch.randint(0, N - 2, (2, E), generator=g)              # the last two nodes keep no edge at all
    ei[:, 5] = ei[:, 4]                                             # a duplicate
    ei[1, 7] = ei[0, 7]                                             # a self loop
    c.edge_index, c.combined_mask, c.n_frag_switch = ei, cm, nfs
    c.t, c.conditions = torch.rand(3, 1, generator=g), torch.rand(3, 1, generator=g)
    return c


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_arbitrary_edge_lists_are_taken_as_given(harness, seed):
    import leftnet_oracle as oracle
    c = _random_graph_case(seed)
    out = run_host(harness, c)
    ref = oracle.dynamics_forward(c.state_dict(torch.float64), c.cfg, [x.double() for x in c.xh], c.edge_index, c.t.double(),
                                  c.conditions.double(), c.n_frag_switch, c.combined_mask, c.cnf, nodeframe="literal")
    v, h = c.split(out)
    rv, rh = c.split(ref)
    print(f"random graph {seed}: vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
    assert rel(v, rv) <= 1e-5 and rel(h, rh) <= 1e-5

