"""The fused training step of DDPMTrainer (oard_loss_prepare / oard_loss_terms / backward sweep / oard_adamw_step: no autograd
graph) against the generic formulation (DiffusionLoss under torch autograd + torch.optim.AdamW) on recorded training steps of the
reference (tests/golden/g9_grad_*.npz): same per-sample nll, same logged terms, same gradient bucket; the AdamW kernel against
torch.optim.AdamW(amsgrad) on random gradients, with and without the clipping factor."""

import pytest
import torch

from _grad_cases import CNF, NODE_NFS, GradCase

pytestmark = pytest.mark.gpu


def _trainer(c, dev, fused, pos_only, **kw):
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.trainer import DDPMTrainer
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=NODE_NFS, edge_nf=0, condition_nf=CNF, device=dev)
    dyn.load_state_dict(c.state_dict(), strict=True)
    return DDPMTrainer(dyn, timesteps=c.meta["T"], norm_values=c.meta["norm_values"], scales=(1.0, 2.0, 1.0), pos_only=pos_only,
                       fused=fused, **kw)


@pytest.mark.parametrize("name,pos_only", [("g9_grad_prod_l2", False), ("g9_grad_h32", True), ("g9_grad_prod_n23", True)])
def test_fused_loss_and_gradients_equal_the_autograd_formulation(name, pos_only):
    c = GradCase(name)                      # g9_grad_prod_l2 holds a t_int = 0 sample: the discretised-likelihood terms are live
    dev = torch.device("cuda:0")
    B = len(c.meta["sizes"])
    t_int = torch.tensor(c.meta["t_int"], dtype=torch.float32, device=dev).view(-1, 1)
    cond = torch.zeros(B, 1, device=dev)

    def draws():
        it = iter(range(c.meta["n_randn"]))
        return lambda shape: torch.from_numpy(c.z[f"randn{next(it)}"]).to(dev)
    # generic: DiffusionLoss + autograd into the bucket
    tg = _trainer(c, dev, False, pos_only)
    tg._bucket.zero_()
    nll_g, info_g = tg.compute_loss((c.reps(torch.float32, dev), cond), training=True, t_int=t_int, draw=draws())
    nll_g.mean(0).backward()
    # fused
    tf = _trainer(c, dev, True, pos_only)
    tf._bucket.zero_()
    nll_f, terms = tf._fused_forward_backward((c.reps(torch.float32, dev), cond), t_int=t_int, draw=draws())
    e_nll = float(((nll_f - nll_g.detach()).abs() / nll_g.detach().abs().clamp(min=1e-6)).max())
    gd = float((tf.flat_grad - tg.flat_grad).norm() / tg.flat_grad.norm())
    gmax = float((tf.flat_grad - tg.flat_grad).abs().max() / tg.flat_grad.abs().max())
    print(f"{name} pos_only={pos_only}: nll rel {e_nll:.2e}; gradient bucket |d|_2/|g|_2 {gd:.2e}, max|d|/max|g| {gmax:.2e}")
    assert e_nll <= 2e-6 and gd <= 2e-6 and gmax <= 2e-6
    K = 3
    for k in range(K):
        assert abs(float(terms[k].mean()) / (tf.loss.scales[k] + 1e-4) - info_g[f"error_t_{k}"]) <= 2e-6 * max(1.0, abs(info_g[f"error_t_{k}"]))
        assert abs(float(terms[K + k].mean()) - info_g[f"unorm_error_t_{k}"]) <= 2e-6 * max(1.0, abs(info_g[f"unorm_error_t_{k}"]))


def test_fused_step_runs_and_moves_the_weights():
    c = GradCase("g9_grad_h32")
    dev = torch.device("cuda:0")
    tr = _trainer(c, dev, True, True)
    w0 = tr.flat_param.clone()
    B = len(c.meta["sizes"])
    batch = (c.reps(torch.float32, dev), torch.zeros(B, 1, device=dev))
    torch.manual_seed(3)
    i1 = tr.training_step(batch)
    i2 = tr.training_step(batch)
    assert i1["skipped"] == 0 and i2["skipped"] == 0 and i1["grad_norm"] > 0
    assert all(k in i1 for k in ("loss", "error_t_0", "unorm_error_t_2", "max_grad_norm"))
    d = (tr.flat_param - w0).abs()
    assert float(d.max()) > 0 and float(d.max()) <= 2.1 * 2.5e-4        # two AdamW steps of lr each, at most
    # the module's parameters ARE the bucket (state_dict sees the update), and the next forward repacks them
    sd = tr.dynamics.state_dict()
    assert any(not torch.equal(sd[k].cpu(), v) for k, v in c.state_dict().items() if v.is_floating_point() and "radial_emb" not in k)
    assert len(tr.gradnorm_queue) == 3


@pytest.mark.parametrize("amsgrad,wd,gscale", [(True, 0.0, 1.0), (True, 0.01, 0.37), (False, 0.0, 1.0)])
def test_adamw_kernel_matches_torch(amsgrad, wd, gscale):
    from oareactdiff_amd import _capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    n = 100003
    p0 = torch.randn(n, generator=g).to(dev)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=2.5e-4, betas=(0.9, 0.999), weight_decay=wd, amsgrad=amsgrad, foreach=False)
    p, m, v, vm = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for step in range(1, 6):
        gr = (torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-3, 2, (1,), generator=g)))).to(dev)
        ref.grad = gr * gscale
        opt.step()
        _capi.check(_capi.lib().oard_adamw_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), vm.data_ptr(), n, 2.5e-4, 0.9, 0.999,
                                                1e-8, wd, step, 1 if amsgrad else 0, gscale, stream), "adamw")
        e = float((p - ref.detach()).abs().max())
        assert e <= 3e-6, (step, e)           # a few ulp of O(1..5) weights (the derived scalars are rounded once, like torch does)


def test_fused_steps_are_deterministic_and_independent_of_the_stream_split():
    """The sweep runs its weight-gradient work on a second stream (forked per stage, per-layer scratch double-buffered).  Every kernel
    is deterministic and every gradient entry is accumulated by the same launches in the same order either way, so three training
    steps from the same seed must give bit-identical weights with the split on (twice) and off."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_training_batch
    from oareactdiff_amd import _capi
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    from oareactdiff_amd.trainer import DDPMTrainer
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)
    batches = [make_training_batch(16, 23, 100 + k, dev) for k in range(2)]
    res = []
    try:
        for dual in (1, 1, 0):
            assert _capi.lib().oard_debug_option(b"train_dual", dual) == 0
            dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
            dyn.load_state_dict(sd, strict=True)
            tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
            torch.manual_seed(1234)
            infos = [tr.training_step(batches[i % 2]) for i in range(3)]
            res.append((tr.flat_param.clone(), tr.flat_grad.clone(), [i_["loss"] for i_ in infos]))
    finally:
        _capi.lib().oard_debug_option(b"train_dual", 1)
    for other in res[1:]:
        assert other[2] == res[0][2]
        assert torch.equal(other[1], res[0][1]) and torch.equal(other[0], res[0][0])
    assert all(torch.isfinite(torch.tensor(res[0][2])))


def test_fused_training_lowers_a_fixed_objective():
    """End to end: with the same t / noise draws every step (the RNG is re-seeded) the objective is a fixed function of the weights;
    60 fused steps (HIP forward, sweep, clipping, AdamW kernel, repacked weights every step) must lower it substantially; so must
    the generic path (autograd through DynamicsFunction + torch.optim.AdamW) on ITS fixed objective (the two paths consume the RNG
    stream differently, so the objectives are different draws of the same loss)."""
    c = GradCase("g9_grad_h32")
    dev = torch.device("cuda:0")
    B = len(c.meta["sizes"])
    batch = (c.reps(torch.float32, dev), torch.zeros(B, 1, device=dev))
    curves = {}
    for fused in (True, False):
        tr = _trainer(c, dev, fused, True)
        for g in tr.optimizer.param_groups:          # the one place the learning rate lives, in both modes
            g["lr"] = 2e-3
        losses = []
        for step in range(60):
            torch.manual_seed(11)
            losses.append(tr.training_step(batch)["loss"])
        curves[fused] = losses
    f, g = curves[True], curves[False]
    print("fixed objective: fused", [round(x, 4) for x in f[::10]], "generic", [round(x, 4) for x in g[::10]])
    assert all(torch.isfinite(torch.tensor(f))) and f[-1] < 0.7 * f[0], (f[0], f[-1])
    assert all(torch.isfinite(torch.tensor(g))) and g[-1] < 0.7 * g[0], (g[0], g[-1])


def test_fused_step_follows_optimizer_param_groups_and_resumes_from_state_dict():
    """Round-3 advisor finding: `trainer.optimizer` is what a caller attaches an LR scheduler to (pl_trainer.py:149-151); the fused
    step reads its hyper-parameters from `optimizer.param_groups[0]` at every step, and `state_dict()` / `load_state_dict()` carry
    the fused AdamW moments, the step counter and the clipping history: a resumed trainer continues bit-identically."""
    c = GradCase("g9_grad_h32")
    dev = torch.device("cuda:0")
    B = len(c.meta["sizes"])
    batch = (c.reps(torch.float32, dev), torch.zeros(B, 1, device=dev))

    def run(tr, steps, seed0):
        out = []
        for k in range(steps):
            torch.manual_seed(seed0 + k)
            out.append(tr.training_step(batch)["loss"])
            tr_sched[id(tr)].step()
        return out
    tr_sched = {}
    a = _trainer(c, dev, True, True)
    tr_sched[id(a)] = torch.optim.lr_scheduler.StepLR(a.optimizer, step_size=2, gamma=0.5)
    w0 = a.flat_param.clone()
    run(a, 2, 0)
    assert a.optimizer.param_groups[0]["lr"] == pytest.approx(0.5 * a.opt_config["lr"])
    # a zero learning rate set through the param group freezes the weights: the fused kernel reads the group, not a private copy
    frozen = _trainer(c, dev, True, True)
    tr_sched[id(frozen)] = torch.optim.lr_scheduler.StepLR(frozen.optimizer, step_size=1000)
    frozen.optimizer.param_groups[0]["lr"] = 0.0
    run(frozen, 2, 0)
    assert torch.equal(frozen.flat_param, w0) and not torch.equal(a.flat_param, w0)
    # resume: module weights + trainer state into a fresh trainer, then both continue identically
    msd = {k: v.clone() for k, v in a.dynamics.state_dict().items()}
    tsd, ssd = a.state_dict(), tr_sched[id(a)].state_dict()
    rest_a = run(a, 3, 100)
    b = _trainer(c, dev, True, True)
    b.dynamics.load_state_dict(msd, strict=True)
    b.load_state_dict(tsd)
    tr_sched[id(b)] = torch.optim.lr_scheduler.StepLR(b.optimizer, step_size=2, gamma=0.5)
    tr_sched[id(b)].load_state_dict(ssd)
    rest_b = run(b, 3, 100)
    assert rest_a == rest_b and torch.equal(a.flat_param, b.flat_param)
    assert torch.equal(a.exp_avg_sq, b.exp_avg_sq) and a.opt_step == b.opt_step == 5
    assert a.gradnorm_queue.items == b.gradnorm_queue.items


def test_step_without_host_sync_takes_the_same_decisions():
    """`host_sync=False` (round 4): the clipping decision (pl_trainer.py:391-418: 1.5 mean + 3 std of the last <= 50 norms), the skip
    decision and AdamW's step-dependent scalars on the device, no device -> host read inside the step.  Against the host-side
    decision over eight steps that contain clipped steps (a small history with more than 8 entries: numpy's pairwise summation
    order), unclipped steps and a skipped step (a NaN position): same losses, same history, same counters, same weights."""
    from oareactdiff_amd.trainer import LazyInfo
    c = GradCase("g9_grad_h32")
    dev = torch.device("cuda:0")
    B = len(c.meta["sizes"])
    good = (c.reps(torch.float32, dev), torch.zeros(B, 1, device=dev))
    bad_reps = c.reps(torch.float32, dev)
    bad_reps[0]["pos"] = bad_reps[0]["pos"].clone()
    bad_reps[0]["pos"][0, 0] = float("nan")
    bad = (bad_reps, torch.zeros(B, 1, device=dev))
    hist = [3e-3 * (1 + 0.1 * k) for k in range(11)]          # far below the real norms: the first steps clip and push max_norm

    def run(host_sync):
        tr = _trainer(c, dev, True, True, host_sync=host_sync)
        tr.gradnorm_queue.items = list(hist)
        infos = []
        for k in range(8):
            torch.manual_seed(50 + k)
            infos.append(tr.training_step(bad if k == 4 else good))
            if k == 2:                                         # a large entry: the following steps are not clipped
                if host_sync:
                    tr.gradnorm_queue.items.insert(0, 1e6)
                else:
                    tr._pull_clip_state(); tr.gradnorm_queue.items.insert(0, 1e6); tr._clip_state = None
        return tr, infos
    a, ia = run(True)
    b, ib = run(False)
    assert all(isinstance(i, LazyInfo) for i in ib) and all(isinstance(i, dict) for i in ia)
    sa, sb = a.state_dict(), b.state_dict()                    # pulls the device-side history and counters
    assert [i["skipped"] for i in ia] == [i["skipped"] for i in ib] == [0, 0, 0, 0, 1, 0, 0, 0]
    assert [i["loss"] for i in ia if i["skipped"] == 0] == [i["loss"] for i in ib if i["skipped"] == 0]
    clipped = [i["grad_norm"] > i["max_grad_norm"] for i in ia if i["skipped"] == 0]
    assert any(clipped) and not all(clipped), clipped
    for x, y in zip(ia, ib):
        assert set(x) == set(y)
        if not x["skipped"]:
            assert x["grad_norm"] == y["grad_norm"] and abs(x["max_grad_norm"] - y["max_grad_norm"]) <= 1e-6 * abs(x["max_grad_norm"])
    assert sa["opt_step"] == sb["opt_step"] == 7 and sa["skipped_steps"] == sb["skipped_steps"] == 1
    assert sa["gradnorm_queue"] == sb["gradnorm_queue"], (sa["gradnorm_queue"][:3], sb["gradnorm_queue"][:3])
    d = float((a.flat_param - b.flat_param).abs().max())
    print(f"host decision vs device decision after 8 steps: max |dw| = {d:.3e} (identical: {torch.equal(a.flat_param, b.flat_param)})")
    assert torch.allclose(a.flat_param, b.flat_param, rtol=1e-6, atol=1e-9)
    assert torch.allclose(a.max_exp_avg_sq, b.max_exp_avg_sq, rtol=1e-6, atol=1e-12)


def test_host_side_layout_and_pooled_topologies_change_nothing():
    """Round 4: a batch moved with DDPMTrainer.to_device carries the host copies of mask / size, the fused step builds its layout from
    them (no device -> host copy) and never materialises an edge list; topologies are created and dropped per step from the library's
    table pool.  Same draws -> bit-identical loss and gradient bucket as the same batch given as plain device tensors; and a run over
    batches of CHANGING sizes (every step a new layout, blocks of different capacity recycled) stays finite and deterministic."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_training_batch
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    from oareactdiff_amd.trainer import DDPMTrainer
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)

    def trainer():
        dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
        dyn.load_state_dict(sd, strict=True)
        return DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True, host_sync=False)
    reps, cond = make_training_batch(6, 11, 77, dev)
    plain = ([{k: v for k, v in r.items() if not k.endswith("_host")} for r in reps], cond)
    ta, tb = trainer(), trainer()
    torch.manual_seed(5)
    ia = ta.training_step((reps, cond))
    torch.manual_seed(5)
    ib = tb.training_step(plain)
    assert ia["loss"] == ib["loss"] and torch.equal(ta.flat_grad, tb.flat_grad) and torch.equal(ta.flat_param, tb.flat_param)
    # changing sizes: 12 steps, each a new layout (atoms per object 5 .. 16, batch 3 .. 8), twice from the same seeds
    runs = []
    for _ in range(2):
        tr = trainer()
        losses = []
        for k in range(12):
            batch = make_training_batch(3 + k % 6, 5 + (7 * k) % 12, 300 + k, dev)
            torch.manual_seed(900 + k)
            losses.append(tr.training_step(batch))
        runs.append(([i["loss"] for i in losses], tr.flat_param.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert all(torch.isfinite(torch.tensor(runs[0][0])))


def test_fused_true_is_refused_where_the_fused_kernels_do_not_apply():
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.trainer import DDPMTrainer
    c = GradCase("g9_grad_h32")
    dev = torch.device("cuda:0")
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=NODE_NFS, edge_nf=0, condition_nf=CNF, device=dev)
    with pytest.raises(ValueError):
        DDPMTrainer(dyn, timesteps=c.meta["T"], loss_type="vlb", fused=True)


@pytest.mark.parametrize("name,pos_only", [("g9_grad_prod_l2", False), ("g9_grad_prod_n23", True)])
def test_two_microbatches_give_the_step_of_one(name, pos_only):
    """`DDPMTrainer(microbatches=2)` (round 6): the step's reactions as two halves on two streams - own tape / workspace / scratch and
    gradient buffer per half, the two sweeps issued step by step - against the one-batch step on the same recorded randomness: the same
    per-sample nll and gradient up to summation order, the same weights after the optimiser step; run twice, the two-stream step is
    bit-identical to itself."""
    from oareactdiff_amd.trainer import DDPMTrainer
    c = GradCase(name)
    dev = torch.device("cuda:0")
    B = len(c.meta["sizes"])
    assert B >= 2
    t_int = torch.tensor(c.meta["t_int"], dtype=torch.float32, device=dev).view(-1, 1)

    def draws():
        it = iter(range(c.meta["n_randn"]))
        return lambda shape: torch.from_numpy(c.z[f"randn{next(it)}"]).to(dev)

    def batch():
        return DDPMTrainer.to_device((c.reps(torch.float32, "cpu"), torch.zeros(B, 1)), dev, non_blocking=False)
    res = {}
    for key, mb in (("one", 1), ("two", 2), ("two again", 2)):
        tr = _trainer(c, dev, True, pos_only, microbatches=mb)
        tr._bucket.zero_()
        nll, terms = tr._fused_forward_backward(batch(), t_int=t_int, draw=draws())
        torch.cuda.synchronize()
        res[key] = (nll.clone(), terms.clone(), tr.flat_grad.clone())
        assert (mb == 2) == (tr._mb is not None)
    # (every reaction's forward is its own, but a half-size launch may pick other node-stage shapes: last bits, not bit-identity)
    assert float(((res["one"][0] - res["two"][0]).abs() / res["one"][0].abs().clamp(min=1e-6)).max()) <= 2e-6
    assert float((res["one"][1] - res["two"][1]).abs().max()) <= 2e-6 * max(1.0, float(res["one"][1].abs().max()))
    g1, g2 = res["one"][2], res["two"][2]
    gd = float((g2 - g1).norm() / g1.norm())
    gmax = float((g2 - g1).abs().max() / g1.abs().max())
    print(f"{name}: two micro-batches vs one: gradient |d|_2/|g|_2 {gd:.2e}, max|d|/max|g| {gmax:.2e}")
    assert gd <= 1e-6 and gmax <= 1e-6
    for a, b in zip(res["two"], res["two again"]):
        assert torch.equal(a, b)
    # whole steps (no host sync, fresh batch objects): the weights after three steps agree
    w = {}
    for mb in (1, 2):
        tr = _trainer(c, dev, True, pos_only, microbatches=mb, host_sync=False)
        for _ in range(3):
            tr.training_step(batch(), t_int=t_int, draw=draws())
        torch.cuda.synchronize()
        w[mb] = tr.flat_param.clone()
    assert float((w[1] - w[2]).abs().max()) <= 1e-6 * float(w[1].abs().max())
