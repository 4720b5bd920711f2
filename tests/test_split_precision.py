"""The optional split-precision edge kernels (GCL and EquiMessage, csrc/oard_edge_b3.h: every fp32 value as three bf16 terms, six bf16 MFMAs per
K block, fp32 accumulation) against the same bar as the fp32 kernel: the reference evaluated in float64, <= 1e-5 of the largest
output, on every golden case in the throughput launch shapes and inside the B = 64 launch bench.py times.  The choice is a module
attribute (`EGNNDynamics.edge_precision`) that travels with every library call as `oard_config.precision` - the library has no
process-wide precision state.  (OARD_GCL_B3=1 OARD_EQUI_B3=1 OARD_TRAIN_B3=1 make it the default of every module whose attribute is
None: the WHOLE GPU suite then runs on these kernels.)"""
import threading

import os

import pytest
import torch

from _cases import ALL_CASES, THROUGHPUT, Case, debug_options, rel
from test_hip_parity import _args, _dyn

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("name", ALL_CASES)
def test_split_precision_forward_matches_reference_f64(name):
    dev = torch.device("cuda:0")
    c = Case(name)
    with debug_options(**THROUGHPUT), torch.no_grad():                   # (the split-precision forms exist for the throughput shapes)
        d32, d3 = _dyn(c, dev), _dyn(c, dev)
        d32.edge_precision, d3.edge_precision = "f32", "bf16x3"
        out32, _ = d32(*_args(c, dev))                                # the fp32 kernels
        out, _ = d3(*_args(c, dev))
    v, h = c.split([o.cpu() for o in out])
    v32, h32 = c.split([o.cpu() for o in out32])
    rv, rh = c.split(c.ref64)
    print(f"{name}: bf16x3-vs-ref64 vel {rel(v, rv):.2e} h {rel(h, rh):.2e} | fp32-vs-ref64 vel {rel(v32, rv):.2e} h {rel(h32, rh):.2e} "
          f"| bf16x3-vs-fp32 vel {rel(v, v32):.2e}")
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL
    if c.cfg["hidden_channels"] >= 32:                              # the kernel really differs from the fp32 one (not bit-identical)
        assert not all(torch.equal(a, b) for a, b in zip(out, out32))


@pytest.mark.parametrize("parts", [0, 1])
def test_split_precision_benched_launch(parts):
    """B = 64 x 3 x 23 atoms, L = 6 with the two reactions of golden g2 in slots 0, 1 and 62, 63 (as tests/test_configs.py)."""
    from oareactdiff_amd.synthetic import make_inputs, make_topology
    from test_configs import _prod_dynamics
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    B, nf = 64, 23
    with debug_options(parts=parts):
        dyn, _, _ = _prod_dynamics(dev, c.cfg)
        dyn.edge_precision = "bf16x3"
        cm, nfs, ei, masks = make_topology(B, nf)
        xh = make_inputs(B, nf, masks, 99, "cpu")
        g = torch.Generator().manual_seed(1)
        t, cond = torch.rand(B, 1, generator=g), torch.rand(B, 1, generator=g)
        for slot0 in (0, 62):
            for k in range(3):
                xh[k][slot0 * nf:(slot0 + 2) * nf] = c.xh[k]
            t[slot0:slot0 + 2], cond[slot0:slot0 + 2] = c.t, c.conditions
        with torch.no_grad():
            out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    rv, rh = c.split(c.ref64)
    for slot0 in (0, 62):
        v, h = c.split([o[slot0 * nf:(slot0 + 2) * nf].cpu() for o in out])
        print(f"parts={parts} slots {slot0},{slot0 + 1}: vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
        assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


def test_edge_precision_attribute_switches_kernels_per_module():
    """`EGNNDynamics.edge_precision`: "bf16x3" / "f32" select the kernels for that module's calls (and repack its weights when the
    choice changes); two modules with different choices can be used alternately."""
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    a, b = _dyn(c, dev), _dyn(c, dev)
    a.edge_precision, b.edge_precision = "bf16x3", "f32"
    with debug_options(**THROUGHPUT), torch.no_grad():
        oa1, _ = a(*_args(c, dev)); ob1, _ = b(*_args(c, dev)); oa2, _ = a(*_args(c, dev)); ob2, _ = b(*_args(c, dev))
        b.edge_precision = "bf16x3"
        ob3, _ = b(*_args(c, dev))
    assert all(torch.equal(x, y) for x, y in zip(oa1, oa2)) and all(torch.equal(x, y) for x, y in zip(ob1, ob2))
    assert not all(torch.equal(x, y) for x, y in zip(oa1, ob1))          # different kernels
    assert all(torch.equal(x, y) for x, y in zip(oa1, ob3))              # same kernels, same weights -> same bits
    rv, rh = c.split(c.ref64)
    for o in (oa1, ob1):
        v, h = c.split([x.cpu() for x in o])
        assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


@pytest.mark.parametrize("name", ["g9_grad_h32", "g9_grad_prod_l2", "g9_grad_prod_n23"])
def test_split_precision_training_forward_gradients(name):
    """The optional split-precision TRAINING-mode forward (tape written by k_gcl_edge_b3 / k_equi_edge_b3<TRAIN>; fp32 backward): the
    training step's loss and gradients against the reference's float64 autograd with the tolerances of tests/test_grad.py."""
    from test_grad import test_hip_training_step_gradients_match_reference_f64 as run_case
    old = os.environ.get("OARD_TRAIN_B3")
    os.environ["OARD_TRAIN_B3"] = "1"             # the default of `train_edge_precision = None`, resolved at every call
    try:
        run_case(name)
    finally:
        if old is None:
            del os.environ["OARD_TRAIN_B3"]
        else:
            os.environ["OARD_TRAIN_B3"] = old


def test_default_none_never_inherits_another_modules_choice(monkeypatch):
    """Round-3 advisor finding: a module left at `edge_precision = None` used to run whatever process-wide flags the last OTHER
    module had set - on a bf16 weight stream it had never packed.  None now resolves to the environment's default at every call."""
    monkeypatch.delenv("OARD_GCL_B3", raising=False)
    monkeypatch.delenv("OARD_EQUI_B3", raising=False)
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    ref32, b, a = _dyn(c, dev), _dyn(c, dev), _dyn(c, dev)
    ref32.edge_precision, a.edge_precision = "f32", "bf16x3"
    with torch.no_grad():
        want, _ = ref32(*_args(c, dev))
        ob1, _ = b(*_args(c, dev))                     # b packs its weights with no bf16 stream
        a(*_args(c, dev))                              # a runs the split-precision kernels
        ob2, _ = b(*_args(c, dev))                     # b must still be the fp32 module
    assert all(torch.equal(x, y) for x, y in zip(want, ob1)) and all(torch.equal(x, y) for x, y in zip(want, ob2))


def test_two_precisions_on_two_threads_and_streams_are_bitwise_reproducible():
    """SURVEY 8(b) "thread-safe for distinct workspaces": an fp32 module and a split-precision module called alternately from two
    host threads, each on its own stream, give exactly the bits of their single-threaded runs."""
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    mods = [_dyn(c, dev), _dyn(c, dev)]
    mods[0].edge_precision, mods[1].edge_precision = "f32", "bf16x3"
    args = [_args(c, dev), _args(c, dev)]
    from oareactdiff_amd import _capi
    from conftest import SUITE_OPTIONS
    for k, v in THROUGHPUT.items():                    # (process-wide debug options, set for the whole test: the workers are threads)
        assert _capi.lib().oard_debug_option(k.encode(), v) == 0
    with torch.no_grad():
        want = [[o.clone() for o in m(*a)[0]] for m, a in zip(mods, args)]
    torch.cuda.synchronize()
    N_CALLS, errors, got = 40, [], [None, None]
    start = threading.Barrier(2)

    def worker(i):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st), torch.no_grad():
                start.wait()
                outs = [[o.clone() for o in mods[i](*args[i])[0]] for _ in range(N_CALLS)]
            st.synchronize()
            got[i] = outs
        except Exception as e:                         # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in THROUGHPUT:
        _capi.lib().oard_debug_option(k.encode(), SUITE_OPTIONS[k])
    assert not errors, errors
    for i in range(2):
        for outs in got[i]:
            assert all(torch.equal(x, y) for x, y in zip(outs, want[i]))
    assert not all(torch.equal(x, y) for x, y in zip(want[0], want[1]))
