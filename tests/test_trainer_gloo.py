"""The training step's data-parallel half on CPU: two gloo ranks, each with its shard of a batch, end up with the
gradients (and, after AdamW, the weights) a single process computes on the whole batch.  The denoiser is the CPU
oracle under torch autograd here (the HIP backward is checked on the GPU in test_grad.py); what is under test is
`DDPMTrainer`: the flat gradient bucket, its single all-reduce, the adaptive clipping and the optimiser step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CFG = dict(pos_require_grad=False, cutoff=10.0, num_layers=1, hidden_channels=32, num_radial=8, in_hidden_channels=8,
           reflect_equiv=True, legacy=True, update=True, pos_grad=False, single_layer_output=True, object_aware=True)
SIZES = [3, 4, 2, 5]
T_INT = [7.0, 0.0, 50.0, 23.0]
#: BASELINE configs[3]'s data-parallel degree: eight reactions (ragged), one per rank at world size 8
SIZES8 = [3, 4, 2, 5, 3, 2, 4, 3]
T_INT8 = [7.0, 0.0, 50.0, 23.0, 1.0, 38.0, 12.0, 49.0]
T = 50


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_dynamics():
    """`oareactdiff_amd.EGNNDynamics` as the parameter container (reference state-dict names) with the CPU oracle as
    its forward - float64, differentiable by torch autograd."""
    import leftnet_oracle as oracle
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import state_spec, synthetic_state_dict

    class OracleDynamics(EGNNDynamics):
        def forward(self, xh, edge_index, t, conditions, n_frag_switch, combined_mask, edge_attr=None):
            sd = dict(self.named_parameters())
            sd.update(dict(self.named_buffers()))
            return oracle.dynamics_forward(sd, CFG, xh, edge_index, t, conditions, n_frag_switch, combined_mask, 1,
                                           nodeframe="exact"), None
    d = OracleDynamics(model_config=dict(CFG), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=torch.device("cpu"))
    d.load_state_dict(synthetic_state_dict(state_spec(CFG, [9, 9, 9], 1), CFG, seed=42))
    return d.double()


def _batch(sizes, lo, hi, t_all=None):
    """Reactions [lo, hi) of the full synthetic batch, plus this shard's rows of the full batch's noise draws."""
    g = torch.Generator().manual_seed(7)
    B = len(sizes)
    full_mask = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    keep = (full_mask >= lo) & (full_mask < hi)
    reps, draws = [], []
    for k in range(3):
        n = full_mask.numel()
        pos = torch.randn(n, 3, generator=g, dtype=torch.float64)
        typ = torch.randint(0, 4, (n,), generator=g)
        one_hot = torch.zeros(n, 5, dtype=torch.long)
        one_hot[torch.arange(n), typ] = 1
        charge = torch.tensor([1, 6, 7, 8])[typ].view(n, 1)
        draws += [torch.randn(n, 3, generator=g, dtype=torch.float64)[keep], torch.randn(n, 6, generator=g, dtype=torch.float64)[keep]]
        reps.append({"size": torch.tensor(sizes[lo:hi]), "pos": pos[keep], "one_hot": one_hot[keep], "charge": charge[keep],
                     "mask": full_mask[keep] - lo})
    t_int = torch.tensor(T_INT if t_all is None else t_all, dtype=torch.float64).view(-1, 1)[lo:hi]
    it = iter(draws)
    return (reps, torch.zeros(hi - lo, 1, dtype=torch.float64)), t_int, (lambda shape: next(it))


def _step(rank, world, perturb=False, poison_rank=None, sizes=None, t_all=None):
    from oareactdiff_amd.shard import shard_range
    from oareactdiff_amd.trainer import DDPMTrainer
    sizes = SIZES if sizes is None else sizes
    lo, hi = shard_range(len(sizes), rank, world)
    dyn = _oracle_dynamics()
    if perturb and rank > 0:                 # replicas that were NOT built identically (different RNG state / only rank 0 loaded)
        with torch.no_grad():
            for p in dyn.parameters():
                p.add_(0.1 * (rank + 1))
    tr = DDPMTrainer(dyn, timesteps=T, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0))
    batch, t_int, draw = _batch(sizes, lo, hi, t_all)
    if poison_rank == rank:                  # this rank's noise is NaN: its loss and gradients are not finite
        inner = draw
        draw = lambda shape: inner(shape) * float("nan")      # noqa: E731
    info = tr.training_step(batch, t_int=t_int, draw=draw)
    return tr, info


def _worker(rank, world, port, q, perturb=False, poison_rank=None, sizes=None, t_all=None):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if world > 2:
        torch.set_num_threads(1)                      # eight ranks on the container's eight cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr, info = _step(rank, world, perturb, poison_rank, sizes, t_all)
        q.put((rank, tr.flat_grad.numpy().copy(), torch.cat([p.detach().reshape(-1) for p in tr.params]).numpy().copy(),
               info["loss"], info["grad_norm"], info["skipped"], len(tr.gradnorm_queue)))   # numpy: tensors would travel as shm handles
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run_two_ranks(**kw):
    return _run_ranks(2, **kw)


def _run_ranks(world, **kw):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=300) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [(r, torch.from_numpy(g), torch.from_numpy(w), l, n, sk, ql) for r, g, w, l, n, sk, ql in out]


def test_two_rank_training_step_equals_single_process():
    single, info1 = _step(0, 1)
    assert single.world == 1 and len(single.names) == len(single.params)
    assert not any(n.startswith(("model.distance_embedding.", "model.last_layer.")) for n in single.names)
    assert float(single.flat_grad.abs().max()) > 0 and info1["grad_norm"] > 0
    w1 = torch.cat([p.detach().reshape(-1) for p in single.params])
    (_, g0, w0, l0, n0, s0, _), (_, g1, wr1, l1, n1, s1, _) = _run_two_ranks()
    assert s0 == 0 and s1 == 0
    assert torch.equal(g0, g1) and torch.equal(w0, wr1)            # the all-reduce leaves the replicas identical
    assert abs(n0 - info1["grad_norm"]) <= 1e-9 * info1["grad_norm"]
    scale = float(single.flat_grad.abs().max())
    assert float((g0 - single.flat_grad).abs().max()) <= 1e-10 * scale   # mean of the shard means == mean over the batch
    assert float((w0 - w1).abs().max()) <= 1e-12
    assert abs(0.5 * (l0 + l1) - info1["loss"]) <= 1e-10 * abs(info1["loss"])


def test_eight_rank_training_step_equals_single_process():
    """BASELINE configs[3] is 8-way data parallel: eight gloo ranks with ONE reaction each (ragged sizes; rank 1's is at t = 0, rank 2's at
    t = T) against one process on the eight.  After the single all-reduce of the flat bucket (sum, / 8) every rank holds the single
    process's gradient, takes the same clipping decision and ends the step with its weights - bit-identical across the eight replicas."""
    single, info1 = _step(0, 1, sizes=SIZES8, t_all=T_INT8)
    w1 = torch.cat([p.detach().reshape(-1) for p in single.params])
    out = _run_ranks(8, sizes=SIZES8, t_all=T_INT8)
    assert [o[0] for o in out] == list(range(8)) and all(o[5] == 0 for o in out)
    g0, w0 = out[0][1], out[0][2]
    for _, g, w, _, n, _, _ in out[1:]:
        assert torch.equal(g, g0) and torch.equal(w, w0)           # replicas identical after the collective
        assert n == out[0][4]
    scale = float(single.flat_grad.abs().max())
    assert float((g0 - single.flat_grad).abs().max()) <= 1e-10 * scale   # mean of eight per-reaction means == mean over the batch
    assert float((w0 - w1).abs().max()) <= 1e-12
    assert abs(out[0][4] - info1["grad_norm"]) <= 1e-9 * info1["grad_norm"]
    assert abs(sum(o[3] for o in out) / 8 - info1["loss"]) <= 1e-10 * abs(info1["loss"])


def test_replicas_built_differently_are_synchronised_at_construction():
    """torch DDP broadcasts rank 0's parameters and buffers when it wraps the module; `DDPMTrainer.__init__` does the same.
    Rank 1 starts from perturbed weights: after construction + one step both ranks hold what a single process computes."""
    single, info1 = _step(0, 1)
    w1 = torch.cat([p.detach().reshape(-1) for p in single.params])
    (_, g0, w0, _, _, _, _), (_, g1, wr1, _, _, _, _) = _run_two_ranks(perturb=True)
    assert torch.equal(g0, g1) and torch.equal(w0, wr1)
    assert float((w0 - w1).abs().max()) <= 1e-12
    assert float((g0 - single.flat_grad).abs().max()) <= 1e-10 * float(single.flat_grad.abs().max())


def test_non_finite_step_is_skipped_on_every_rank():
    """One rank sees NaN: no rank leaves before the collective (nobody hangs), every rank skips the optimiser step, the
    weights stay what they were and nothing non-finite enters the clipping history."""
    fresh = torch.cat([p.detach().reshape(-1) for p in _oracle_dynamics().parameters() if p.requires_grad])
    (_, g0, w0, l0, n0, s0, q0), (_, g1, wr1, l1, n1, s1, q1) = _run_two_ranks(poison_rank=1)
    assert s0 == 1 and s1 == 1
    assert q0 == 1 and q1 == 1                                   # only the initial 3000 (pl_trainer.py:143-146)
    assert torch.equal(w0, wr1) and bool(torch.isfinite(w0).all())
    from oareactdiff_amd.trainer import DDPMTrainer
    tr = DDPMTrainer(_oracle_dynamics(), timesteps=T)
    assert torch.equal(w0, torch.cat([p.detach().reshape(-1) for p in tr.params]))      # untouched weights
    assert fresh.numel() >= w0.numel()


def test_gradient_clipping_follows_the_reference_rule():
    from oareactdiff_amd.trainer import DDPMTrainer, Queue
    q = Queue(max_len=3)
    for v in (1.0, 2.0, 3.0, 4.0):
        q.add(v)
    assert q.items == [4.0, 3.0, 2.0] and abs(q.mean() - 3.0) < 1e-12
    tr = DDPMTrainer(_oracle_dynamics(), timesteps=T)
    tr.gradnorm_queue = Queue()
    tr.gradnorm_queue.add(1e-3)                                   # allows 1.5e-3: the step's gradient must be clipped
    tr.flat_grad.fill_(1.0)
    norm, allowed = tr.clip_gradients()
    assert abs(allowed - 1.5e-3) < 1e-12 and norm > allowed
    assert abs(float(torch.linalg.vector_norm(tr.flat_grad)) - allowed) <= 1e-6 * allowed
    assert tr.gradnorm_queue.items[0] == allowed                 # the clipped value enters the history (pl_trainer.py:409-412)


def test_trainer_state_dict_resumes_the_generic_path():
    """`DDPMTrainer.state_dict()` / `load_state_dict()` on the generic (autograd + torch.optim.AdamW) path: optimiser state, clipping
    history and skipped-step counter survive a save / restore into a freshly built trainer; the continuation is bit-identical.
    `fused=True` is refused off the HIP module (the fused kernels implement the float32 l2 objective only)."""
    import pytest
    from oareactdiff_amd.trainer import DDPMTrainer

    def steps(tr, n):
        out = []
        for _ in range(n):
            batch, t_int, draw = _batch(SIZES, 0, len(SIZES))
            out.append(tr.training_step(batch, t_int=t_int, draw=draw)["loss"])
        return out
    a = DDPMTrainer(_oracle_dynamics(), timesteps=T, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0))
    assert not a.fused
    steps(a, 2)
    msd = {k: v.clone() for k, v in a.dynamics.state_dict().items()}
    tsd = a.state_dict()
    rest_a = steps(a, 2)
    b = DDPMTrainer(_oracle_dynamics(), timesteps=T, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0))
    b.dynamics.load_state_dict(msd, strict=True)
    b.load_state_dict(tsd)
    assert b.gradnorm_queue.items == tsd["gradnorm_queue"] and len(b.gradnorm_queue) == 3
    rest_b = steps(b, 2)
    assert rest_a == rest_b
    for p, q in zip(a.params, b.params):
        assert torch.equal(p, q)
    with pytest.raises(ValueError):
        DDPMTrainer(_oracle_dynamics(), timesteps=T, fused=True)
    with pytest.raises(ValueError):
        b.load_state_dict(dict(tsd, names=["x"]))
