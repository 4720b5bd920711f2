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
