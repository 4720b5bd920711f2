"""Synthetic workloads of the shape BASELINE.json names (no dataset / checkpoint is available):
fixed-atom-count R/TS/P triples with per-(object, sample) centre-of-mass-free N(0,1) positions and
[one-hot(5) | atomic number] features, as `utils/sampling_tools.py:64-108` +
`diffusion/_utils.py:22-31` of the reference would hand to the sampler."""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import Tensor

from .graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch


def make_topology(batch: int, n_atoms: int, n_obj: int = 3, device="cpu") -> Tuple[Tensor, Tensor, Tensor, List[Tensor]]:
    natm = [torch.full((batch,), n_atoms, dtype=torch.long, device=device) for _ in range(n_obj)]
    masks = [get_mask_for_frag(n) for n in natm]
    combined_mask = torch.cat(masks)
    n_frag_switch = get_n_frag_switch(natm)
    edge_index = get_edges_index(combined_mask, remove_self_edge=True)
    return combined_mask, n_frag_switch, edge_index, masks


def make_inputs(batch: int, n_atoms: int, masks: List[Tensor], seed: int, device, pos_scale: float = 1.0) -> List[Tensor]:
    g = torch.Generator(device="cpu").manual_seed(seed)
    xh = []
    for m in masks:
        n = batch * n_atoms
        pos = torch.randn(n, 3, generator=g)
        mean = torch.zeros(batch, 3).index_add_(0, m.cpu(), pos) / n_atoms
        pos = (pos - mean[m.cpu()]) * pos_scale
        typ = torch.multinomial(torch.tensor([0.5, 0.3, 0.1, 0.1]), n, replacement=True, generator=g)
        z = torch.tensor([1.0, 6.0, 7.0, 8.0])[typ]
        feat = torch.zeros(n, 6)
        feat[torch.arange(n), typ] = 1.0
        feat[:, 5] = z
        xh.append(torch.cat([pos, feat], dim=1).to(device))
    return xh
