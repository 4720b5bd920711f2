"""The optional split-precision edge kernels (GCL and EquiMessage, csrc/oard_edge_b3.h: every fp32 value as three bf16 terms, six bf16 MFMAs per
K block, fp32 accumulation) against the same bar as the fp32 kernel: the reference evaluated in float64, <= 1e-5 of the largest
output, on every golden case in the throughput launch shapes and inside the B = 64 launch bench.py times.  The option is read when
the weights are packed, so every case builds a fresh module inside the option's scope.  (OARD_GCL_B3=1 runs the WHOLE GPU suite on
this kernel.)"""
import pytest
import torch

from _cases import ALL_CASES, Case, debug_options, rel
from test_hip_parity import _args, _dyn

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("name", ALL_CASES)
def test_split_precision_forward_matches_reference_f64(name):
    dev = torch.device("cuda:0")
    c = Case(name)
    with torch.no_grad():
        with debug_options(gcl_b3=0, equi_b3=0):
            out32, _ = _dyn(c, dev)(*_args(c, dev))                   # the fp32 kernels
        with debug_options(gcl_b3=1, equi_b3=1):
            out, _ = _dyn(c, dev)(*_args(c, dev))
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
    with debug_options(parts=parts, gcl_b3=1, equi_b3=1):
        dyn, _, _ = _prod_dynamics(dev, c.cfg)
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
    from oareactdiff_amd import _capi
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    a, b = _dyn(c, dev), _dyn(c, dev)
    a.edge_precision, b.edge_precision = "bf16x3", "f32"
    try:
        with torch.no_grad():
            oa1, _ = a(*_args(c, dev)); ob1, _ = b(*_args(c, dev)); oa2, _ = a(*_args(c, dev)); ob2, _ = b(*_args(c, dev))
            b.edge_precision = "bf16x3"
            ob3, _ = b(*_args(c, dev))
    finally:
        _capi.lib().oard_debug_option(b"gcl_b3", 0)
        _capi.lib().oard_debug_option(b"equi_b3", 0)
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
    with debug_options(train_b3=1):
        run_case(name)
