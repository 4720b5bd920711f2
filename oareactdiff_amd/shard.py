"""Sharding of independent reactions over the GPUs of a node.

Reactions never interact (edges exist only inside one `combined_mask` sample,
oa_reactdiff/utils/_graph_tools.py:30), so the denoising path shards by reaction with no data-path
collective: rank r takes a contiguous slice of the batch, builds its own topology and runs its own
replica of the weights.  Only the wall-clock of a timed region is reduced (MAX) across ranks."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of `total` items for `rank`; sizes differ by at most one, earlier ranks get the extra."""
    assert 0 <= rank < world
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(sizes: Sequence[int], rank: int, world: int) -> List[int]:
    """Per-reaction atom counts of this rank's slice."""
    lo, hi = shard_range(len(sizes), rank, world)
    return list(sizes[lo:hi])


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """MAX of a local wall-clock over all ranks (what bench.py reports); identity without a process group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
