"""Minimal stand-in for the `torch_scatter` package (absent from this image).

TEST INFRASTRUCTURE ONLY: lets `oracle/make_goldens.py` import the reference
(`/root/reference/oa_reactdiff`) in the build container so that golden vectors
can be generated.  Semantics follow the published torch_scatter API for the
three entry points the reference's hot path uses (scatter / scatter_add /
scatter_mean along `dim`, index broadcast, `dim_size` inference).
"""
from typing import Optional

import torch


def _broadcast(index: torch.Tensor, src: torch.Tensor, dim: int) -> torch.Tensor:
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(0, dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def scatter_add(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None):
    index = _broadcast(index, src, dim)
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif index.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(index.max()) + 1
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


scatter_sum = scatter_add


def scatter_mean(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None):
    out = scatter_add(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_add(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count = _broadcast(count, out, dim)
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode="floor")
    return out


def scatter(src, index, dim: int = -1, out=None, dim_size: Optional[int] = None,
            reduce: str = "sum"):
    if reduce in ("sum", "add"):
        return scatter_add(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    raise NotImplementedError(reduce)
