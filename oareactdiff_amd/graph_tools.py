"""Host-side topology helpers with the reference's names and semantics
(oa_reactdiff/utils/_graph_tools.py:9-96).  They build the tensors the diffusion
sampler hands to `EGNNDynamics.forward` once per `sample()`; they are not on the
per-step path.  `get_edges_index` avoids the reference's dense [N, N] adjacency
(1.2 GB at B=512) by enumerating pairs per sample; the result is identical,
including the (row, col) lexicographic order `torch.where(adj)` produces.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor


def get_mask_for_frag(natm: Tensor) -> Tensor:
    """Tensor([2, 0, 3]) -> [0, 0, 2, 2, 2]  (_graph_tools.py:84-96)."""
    return torch.repeat_interleave(torch.arange(natm.size(0), device=natm.device), natm).to(natm.device)


def get_n_frag_switch(natm_list: List[Tensor]) -> Tensor:
    """[Tensor(1, 1), Tensor(2, 1)] -> [0, 0, 1, 1, 1]  (_graph_tools.py:62-81)."""
    shapes = [natm.shape[0] for natm in natm_list]
    assert len(set(shapes)) <= 1, "Tensor must be the same length for <natom_list>"
    dev = natm_list[0].device
    return torch.repeat_interleave(
        torch.arange(len(natm_list), device=dev),
        torch.tensor([int(torch.sum(natm).item()) for natm in natm_list], device=dev),
    ).to(dev)


def get_edges_index(combined_mask: Tensor, pos: Optional[Tensor] = None,
                    edge_cutoff: Optional[float] = None, remove_self_edge: bool = False) -> Tensor:
    """[2, n_edges]: every ordered pair of nodes with equal `combined_mask`
    (_graph_tools.py:9-36), sorted by (row, col)."""
    if edge_cutoff is not None:
        adj = combined_mask[:, None] == combined_mask[None, :]
        adj = adj & (torch.cdist(pos, pos) <= edge_cutoff)
        if remove_self_edge:
            adj = adj.fill_diagonal_(False)
        return torch.stack(torch.where(adj), dim=0)
    dev = combined_mask.device
    cm = combined_mask.detach().cpu()
    n = cm.numel()
    order = torch.argsort(cm, stable=True)                # nodes grouped by sample, ascending id inside
    sorted_cm = cm[order]
    uniq, counts = torch.unique_consecutive(sorted_cm, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    grp_of_sorted = torch.repeat_interleave(torch.arange(uniq.numel()), counts)
    grp = torch.empty(n, dtype=torch.long)
    grp[order] = grp_of_sorted
    deg = counts[grp]                                     # members of own sample (incl. self)
    row = torch.repeat_interleave(torch.arange(n), deg)
    off = torch.arange(row.numel()) - torch.repeat_interleave(torch.cumsum(deg, 0) - deg, deg)
    col = order[starts[grp[row]] + off]
    if remove_self_edge:
        keep = row != col
        row, col = row[keep], col[keep]
    return torch.stack([row, col], dim=0).to(dev)


def get_subgraph_mask(edge_index: Tensor, n_frag_switch: Tensor) -> Tensor:
    """1 for inner-object edges, 0 for inter-object edges  (_graph_tools.py:39-59)."""
    return (n_frag_switch[edge_index[0]] == n_frag_switch[edge_index[1]]).long()


def get_inner_edge_index(subgraph_mask: Tensor) -> Tensor:
    return torch.stack(torch.where(subgraph_mask), dim=0)
