"""The N > 1 path: reactions shard by rank with no data-path collective; only the timed region's wall
clock is MAX-reduced.  Exercised with two gloo processes on CPU (the GPU run uses the same code with
the nccl/RCCL backend)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oareactdiff_amd.shard import max_over_ranks, shard_range, shard_sizes


def test_shard_range_partitions():
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                assert 0 <= lo <= hi <= total and hi - lo in (total // world, total // world + 1)
                cover += list(range(lo, hi))
            assert cover == list(range(total))
    assert shard_sizes([3, 4, 5, 6, 7], 1, 2) == [6, 7]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oareactdiff_amd.synthetic import make_inputs, make_topology
        total = 5                                            # reactions in the whole job
        lo, hi = shard_range(total, rank, world)
        cm, nfs, ei, masks = make_topology(hi - lo, 4)
        xh = make_inputs(hi - lo, 4, masks, 100 + rank, "cpu")
        n_nodes = torch.tensor([cm.numel()], dtype=torch.int64)
        dist.all_reduce(n_nodes)                              # bookkeeping only: every reaction is owned once
        t = max_over_ranks(0.25 * (rank + 1), dist)
        q.put((rank, lo, hi, int(n_nodes.item()), t, xh[0].shape[0], ei.shape[1]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _worker_sampler(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oareactdiff_amd.sampler import DiffusionSampler

        class Stub(DiffusionSampler):                        # the sharding logic only: sample() itself needs a GPU
            def __init__(self):
                self.node_nfs = [9, 9, 9]

            def sample(self, n, frags, conditions=None, h0=None, **kw):
                seed = torch.initial_seed()
                outs = [torch.full((int(f.sum()), 9), float(seed)) for f in frags]
                assert conditions.shape[0] == n and all(h.shape[0] == int(f.sum()) for h, f in zip(h0, frags))
                return [outs], [torch.repeat_interleave(torch.arange(n), f) for f in frags]
        sizes = torch.tensor([3, 4, 2, 5, 6])
        frags = [sizes, sizes + 1, sizes]
        h0 = [torch.arange(int(f.sum())).float().view(-1, 1).repeat(1, 6) for f in frags]
        out, masks, (lo, hi) = Stub().sample_sharded(frags, conditions=torch.zeros(5, 1), seed=100, gather=True, h0=h0)
        q.put((rank, lo, hi, [tuple(o.shape) for o in out[0]], [float(o[0, 0]) for o in out[0]], [float(o[-1, 0]) for o in out[0]]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_sample_sharded_splits_the_global_batch_and_gathers_on_rank0():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sampler, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, sh0, first0, last0), (r1, lo1, hi1, sh1, first1, last1) = out
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)
    assert sh0 == [(20, 9), (25, 9), (20, 9)]                 # rank 0 holds the gathered batch (all 5 reactions)
    assert sh1 == [(11, 9), (13, 9), (11, 9)]                 # rank 1 keeps its own slice
    assert first0 == [100.0] * 3 and last0 == [101.0] * 3     # seed + rank, slices in batch order


def test_two_rank_replicas_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, n0, t0, a0, e0), (r1, lo1, hi1, n1, t1, a1, e1) = out
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)
    assert n0 == n1 == 5 * 3 * 4                              # all reactions covered exactly once
    assert t0 == t1 == 0.5                                    # MAX over ranks
    assert (a0, a1) == (12, 8) and (e0, e1) == (3 * 12 * 11, 2 * 12 * 11)
