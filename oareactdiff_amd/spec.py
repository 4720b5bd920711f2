"""State-dict layout of the hot path and a platform-independent synthetic initialiser.

The names and shapes are exactly those of the reference's `EGNNDynamics(model=LEFTNet)`
state dict (oa_reactdiff/dynamics/_base.py:82-132, oa_reactdiff/model/leftnet.py:594-688),
so a real checkpoint's `ddpm.dynamics.*` tensors load unchanged.  The synthetic
initialiser is an integer-hash generator (no torch RNG) so that the build container, the
GPU box and the committed golden fixtures all see bit-identical weights without any
weight file travelling.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

# kind: "w" weight [out,in], "b" bias, "ln_w", "ln_b", "buf_means", "buf_betas"
SpecEntry = Tuple[Tuple[int, ...], str, int]


def model_dims(model_config: Dict) -> Tuple[int, int, int, int]:
    H = int(model_config.get("hidden_channels", 128))
    R = int(model_config.get("num_radial", 96))
    L = int(model_config.get("num_layers", 4))
    C = int(model_config.get("in_hidden_channels", 8))
    return H, R, L, C


def state_spec(model_config: Dict, node_nfs: List[int], condition_nf: int, pos_dim: int = 3,
               condition_time: bool = True, n_encoders: Optional[int] = None) -> "OrderedDict[str, SpecEntry]":
    H, R, L, C = model_dims(model_config)
    W = 3 * H + R
    spec: "OrderedDict[str, SpecEntry]" = OrderedDict()

    def lin(name: str, out: int, inn: int, bias: bool = True) -> None:
        spec[name + ".weight"] = ((out, inn), "w", inn)
        if bias:
            spec[name + ".bias"] = ((out,), "b", inn)

    def ln(name: str) -> None:
        spec[name + ".weight"] = ((H,), "ln_w", H)
        spec[name + ".bias"] = ((H,), "ln_b", H)

    m = "model."
    lin(m + "embedding", H, C)
    lin(m + "embedding_out", C, H)
    spec[m + "radial_emb.means"] = ((R,), "buf_means", R)
    spec[m + "radial_emb.betas"] = ((R,), "buf_betas", R)
    lin(m + "neighbor_emb.embedding", H, C)
    lin(m + "s2v.lin1.0", H, H)
    lin(m + "radial_lin.0", H, R)
    lin(m + "radial_lin.2", H, H)
    lin(m + "lin3.0", H // 4, 3)
    lin(m + "lin3.2", 1, H // 4)
    lin(m + "pos_expansion.mlp.0.linear", H // 2, 3, bias=False)
    lin(m + "pos_expansion.mlp.1.linear", H, H // 2, bias=False)
    lin(m + "distance_embedding.mlp.0.linear", H // 2, R, bias=False)   # unused in forward
    lin(m + "distance_embedding.mlp.1.linear", H, H // 2, bias=False)   # unused in forward
    for l in range(L):
        g = m + f"gcl_layers.{l}."
        lin(g + "edge_mlp.mlp.0.linear", H, 2 * H + W)
        lin(g + "edge_mlp.mlp.1.linear", H, H)
        lin(g + "node_mlp.mlp.0.linear", H, 2 * H)
        lin(g + "node_mlp.mlp.1.linear", H, H)
        lin(g + "edge_out_trans.mlp.0.linear", W, H)
        lin(g + "att_mlp.mlp.0.linear", 1, H)
        ln(g + "x_layernorm")
    for l in range(L):
        q = m + f"message_layers.{l}."
        lin(q + "dir_proj.0", 3 * H, W)
        lin(q + "dir_proj.2", 3 * H, 3 * H)
        lin(q + "x_proj.0", H, H, bias=False)
        lin(q + "x_proj.2", 3 * H, H, bias=False)
        lin(q + "rbf_proj", 3 * H, R, bias=False)
        ln(q + "x_layernorm")
    for l in range(L):
        u = m + f"update_layers.{l}."
        lin(u + "vec_proj", 2 * H, H, bias=False)
        lin(u + "xvec_proj.0", H, 2 * H, bias=False)
        lin(u + "xvec_proj.2", 3 * H, H, bias=False)
        lin(u + "lin3.0", 48, 3)
        lin(u + "lin3.2", 8, 48)
        lin(u + "lin3.4", 1, 8)
    lin(m + "last_layer", 1, H)                                          # unused in forward
    o = m + "out_pos.output_network.0."
    lin(o + "vec1_proj", H, H, bias=False)
    lin(o + "vec2_proj", 1, H, bias=False)
    lin(o + "update_net.0", H, 2 * H)
    lin(o + "update_net.2", 2, H)

    embed_dim = C - (1 if condition_time else 0) - (condition_nf if condition_nf > 0 else 0)
    assert embed_dim > 0
    n_obj = len(node_nfs)
    for prefix, enc in (("encoders", True), ("decoders", False)):
        for k in range(n_obj):
            d = node_nfs[k] - pos_dim
            if enc:
                lin(f"{prefix}.{k}.mlp.0.linear", 2 * d, d)
                lin(f"{prefix}.{k}.mlp.1.linear", embed_dim, 2 * d)
            else:
                lin(f"{prefix}.{k}.mlp.0.linear", 2 * d, embed_dim)
                lin(f"{prefix}.{k}.mlp.1.linear", d, 2 * d)
    return spec


# ---- integer-hash uniform generator ----------------------------------------------------------
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for c in s.encode("utf-8"):
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [0,1) with 24 significant bits; depends only on (key, seed, index)."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base) & _MASK
    z = _splitmix64(idx)
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def hash_normal(key: str, n: int, seed: int = 0) -> np.ndarray:
    """Box-Muller on two hash_uniform streams (float64)."""
    u1 = hash_uniform(key + "#u1", n, seed)
    u2 = hash_uniform(key + "#u2", n, seed)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def rbf_buffers(num_radial: int, cutoff: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """`RBFEmb._initial_params` (leftnet.py:49-56): float32 buffers."""
    start = torch.exp(torch.scalar_tensor(-float(cutoff)))
    end = torch.exp(torch.scalar_tensor(-0.0))
    means = torch.linspace(start, end, num_radial)
    betas = torch.tensor([(2 / num_radial * (end - start)) ** -2] * num_radial)
    return means, betas


def synthetic_state_dict(spec: "OrderedDict[str, SpecEntry]", model_config: Dict, seed: int = 42,
                         dtype: torch.dtype = torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights and biases (PyTorch-default-like scale);
    LayerNorm gamma = 1 + 0.2*(u-0.5), beta = 0.2*(u-0.5) so the affine part is exercised."""
    H, R, L, C = model_dims(model_config)
    means, betas = rbf_buffers(R, float(model_config.get("cutoff", 10.0)))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, (shape, kind, fan_in) in spec.items():
        n = int(np.prod(shape))
        if kind in ("w", "b"):
            bound = 1.0 / math.sqrt(fan_in)
            v = (hash_uniform(name, n, seed) * 2.0 - 1.0) * bound
        elif kind == "ln_w":
            v = 1.0 + 0.2 * (hash_uniform(name, n, seed) - 0.5)
        elif kind == "ln_b":
            v = 0.2 * (hash_uniform(name, n, seed) - 0.5)
        elif kind == "buf_means":
            sd[name] = means.to(dtype)
            continue
        elif kind == "buf_betas":
            sd[name] = betas.to(dtype)
            continue
        else:
            raise ValueError(kind)
        # weights are defined as float32 values (what a checkpoint holds), then cast
        sd[name] = torch.from_numpy(v.astype(np.float32).reshape(shape)).to(dtype)
    return sd


PRODUCTION_LEFTNET_CONFIG = dict(  # oa_reactdiff/trainer/train_ts1x.py:43-56
    pos_require_grad=False, cutoff=10.0, num_layers=6, hidden_channels=196, num_radial=96,
    in_hidden_channels=8, reflect_equiv=True, legacy=True, update=True, pos_grad=False,
    single_layer_output=True, object_aware=True,
)
