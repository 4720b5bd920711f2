"""CPU oracle for the OA-ReactDiff denoising hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (state-dict in, tensors out) restatement
of what the reference computes in

    oa_reactdiff/dynamics/egnn_dynamics.py:63-168   EGNNDynamics.forward
    oa_reactdiff/dynamics/_base.py:82-132           encoders / decoders
    oa_reactdiff/model/leftnet.py:724-891           LEFTNet.forward
    oa_reactdiff/model/core.py:52-92                MLP
    oa_reactdiff/model/util_funcs.py:27-45          unsorted_segment_sum
    oa_reactdiff/utils/_graph_tools.py:9-96         topology builders

It exists to CHECK the HIP path (tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg).  Nothing in `oareactdiff_amd/` may import it; the product
path has no CPU fallback.

Pinning: `oracle/make_goldens.py` (run in the build container, where the
reference is importable through `oracle/_stubs`) checks this restatement against
the imported reference in float64 (<= 1e-10, literal node-frame arithmetic) and
writes the fixtures under `tests/golden/`; `tests/test_oracle_golden.py` re-checks
the oracle against those committed fixtures without the reference.

Third-party arithmetic on this path that is not in /root/reference:
torch_scatter (unpinned, env.yaml:17) and torch_geometric MessagePassing
(unpinned, env.yaml:19).  Their semantics here are index-gather + scatter
sum/mean; they are restated inline (`_scatter_sum`, `_scatter_mean`).

All functions work in the dtype of the inputs (float32 or float64).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

EPS = 1e-6  # leftnet.py:15


# --------------------------------------------------------------------------------------
# topology (utils/_graph_tools.py)
# --------------------------------------------------------------------------------------
def get_mask_for_frag(natm: Tensor) -> Tensor:
    """_graph_tools.py:84-96 — sample index of every node of one object."""
    return torch.repeat_interleave(torch.arange(natm.size(0)), natm)


def get_n_frag_switch(natm_list: Sequence[Tensor]) -> Tensor:
    """_graph_tools.py:62-81 — object index of every node."""
    return torch.repeat_interleave(
        torch.arange(len(natm_list)),
        torch.tensor([int(n.sum()) for n in natm_list]),
    )


def get_edges_index(combined_mask: Tensor, remove_self_edge: bool = True) -> Tensor:
    """_graph_tools.py:9-36 — complete graph per sample, (row, col) lexicographic."""
    adj = combined_mask[:, None] == combined_mask[None, :]
    if remove_self_edge:
        adj = adj.clone()
        adj.fill_diagonal_(False)
    return torch.stack(torch.where(adj), dim=0)


def get_subgraph_mask(edge_index: Tensor, n_frag_switch: Tensor) -> Tensor:
    """_graph_tools.py:39-59 — 1 for an edge inside one object, 0 across objects."""
    return (n_frag_switch[edge_index[0]] == n_frag_switch[edge_index[1]]).long()


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def _linear(x: Tensor, sd: Dict[str, Tensor], name: str) -> Tensor:
    w = sd[name + ".weight"]
    y = x @ w.t()
    b = sd.get(name + ".bias")
    if b is not None:
        y = y + b
    return y


def _layer_norm(x: Tensor, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None,
                eps: float = 1e-5) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y


def _mlp(x: Tensor, sd: Dict[str, Tensor], prefix: str, n: int,
         last_layer_no_activation: bool = False) -> Tensor:
    """core.py:52-92 — `n` x (Linear, SiLU); optionally no activation on the last."""
    for k in range(n):
        x = _linear(x, sd, f"{prefix}.mlp.{k}.linear")
        if not (last_layer_no_activation and k == n - 1):
            x = _silu(x)
    return x


def _scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def _scatter_mean(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    s = _scatter_sum(src, index, dim_size)
    cnt = torch.zeros(dim_size, dtype=src.dtype).index_add_(
        0, index, torch.ones(index.numel(), dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return s / cnt.view((-1,) + (1,) * (src.dim() - 1))


def assemble_nodemask(edge_index: Tensor, n_nodes: int) -> Tensor:
    """leftnet.py:707-722 — one-hop labelling with overwrites, in node order."""
    labels = [-1] * n_nodes
    nbrs: List[List[int]] = [[] for _ in range(n_nodes)]
    for a, b in zip(edge_index[0].tolist(), edge_index[1].tolist()):
        nbrs[a].append(b)
    ind = 0
    for center in range(n_nodes):
        if labels[center] > -1:
            continue
        for j in nbrs[center]:
            labels[j] = ind
        labels[center] = ind
        ind += 1
    return torch.tensor(labels, dtype=torch.long)


def rbf_emb(dist: Tensor, means: Tensor, betas: Tensor, cutoff: float) -> Tensor:
    """leftnet.py:63-69 — exponential-normal radial basis with cosine envelope.
    `means` / `betas` are the module's registered buffers (state-dict entries
    `model.radial_emb.{means,betas}`, built in float32 by leftnet.py:49-56)."""
    dt = dist.dtype
    d = dist.unsqueeze(-1)
    rbounds = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0)
    rbounds = rbounds * (d < cutoff).to(dt)
    return rbounds * torch.exp(-betas.to(dt) * torch.square(torch.exp(-d) - means.to(dt)))


# --------------------------------------------------------------------------------------
# LEFTNet.forward (leftnet.py:724-891)
# --------------------------------------------------------------------------------------
def leftnet_forward(
    sd: Dict[str, Tensor],
    cfg: Dict,
    h: Tensor,
    pos: Tensor,
    edge_index: Tensor,
    subgraph_mask: Optional[Tensor],
    nodeframe: str = "literal",
    prefix: str = "model.",
    stages: Optional[Dict[str, Tensor]] = None,
    geom64: bool = False,
) -> Tuple[Tensor, Tensor]:
    """Returns (h_out [N, in_hidden], dpos [N, 3]).

    `nodeframe`:
      "literal" — the reference's arithmetic: b = mean of neighbours' pos_frame,
                  y1 = a x b / (|a x b| + eps)  (leftnet.py:812-829).
      "exact"   — the exact-arithmetic value for complete-per-sample graphs:
                  sum_sample(pos_frame) == 0  =>  b = -a/(n_s-1), a x b == 0,
                  y1 = z1 = 0.  This is what the HIP path
                  implements; in float64 the two agree to ~1e-9.
    `geom64`: evaluate the geometry block (distances, cutoff mask, labels, frame-CoM
    removal, edge frames, RBF, node frame) in float64 from the given positions and cast
    the results to the working dtype.  The float32 reference is ill-conditioned there
    (cross products of (anti)parallel vectors divided by |.|+1e-6); float64 geometry makes a
    float32 network track the float64 reference.  This models what the HIP path does.
    Only the production switches are restated: legacy=True, update=True,
    pos_grad=False, single_layer_output=True, object_aware=True, for_conf=False.
    """
    H = cfg["hidden_channels"]
    R = cfg["num_radial"]
    L = cfg["num_layers"]
    cutoff = float(cfg["cutoff"])
    reflect_equiv = cfg.get("reflect_equiv", True)
    dt = pos.dtype
    N = pos.size(0)
    p = prefix
    st = stages if stages is not None else {}

    i, j = edge_index[0], edge_index[1]                                   # :741

    z_emb = _linear(h, sd, p + "embedding")                               # :744

    # ---- geometry block, evaluated in `gdt` (float64 when geom64) then cast to `dt` ----------
    gdt = torch.float64 if geom64 else dt
    gpos = pos.to(gdt)
    dist0 = (gpos[i] - gpos[j]).pow(2).sum(dim=-1).sqrt()                 # :747
    mask = (dist0 < cutoff).to(gdt).unsqueeze(-1)                         # :748-751
    if subgraph_mask is not None:
        mask = mask * subgraph_mask.to(gdt).view(-1, 1)                   # :752-753

    ei_cut = edge_index[:, mask.squeeze(-1) > 0]                          # :755
    labels = assemble_nodemask(ei_cut, N)                                 # :756-758
    st["labels"] = labels
    pos_frame = gpos - _scatter_mean(gpos, labels, int(labels.max()) + 1 if N else 0)[labels]  # :760-761

    # scalarization (leftnet.py:693-705); torch.cross there has no `dim` and takes the
    # first size-3 axis — identical to dim=1 unless E == 3, which callers avoid.
    diff = pos_frame[i] - pos_frame[j]
    dist = diff.pow(2).sum(dim=-1).sqrt()
    radial = (diff ** 2).sum(1, keepdim=True)
    cross = torch.linalg.cross(pos_frame[i], pos_frame[j], dim=1)
    coord_diff = diff / (torch.sqrt(radial) + EPS)
    coord_cross = cross / (torch.sqrt((cross ** 2).sum(1, keepdim=True)) + EPS)
    coord_vertical = torch.linalg.cross(coord_diff, coord_cross, dim=1)

    dist = dist * mask.squeeze(-1)                                        # :768
    coord_diff = coord_diff * mask                                        # :769
    coord_cross = coord_cross * mask                                      # :770
    coord_vertical = coord_vertical * mask                                # :771
    radial_emb = rbf_emb(dist, sd[p + "radial_emb.means"], sd[p + "radial_emb.betas"], cutoff) * mask  # :781-782
    rbounds = 0.5 * (torch.cos(dist * math.pi / cutoff) + 1.0)            # :785

    # node frame (leftnet.py:812-834)
    a = pos_frame
    if nodeframe == "literal":
        b = _scatter_mean(pos_frame[i], j, N)                             # vector(): mean at edge_index[1]
        x1 = (a - b) / (torch.sqrt(((a - b) ** 2).sum(1, keepdim=True)) + EPS)
        y1 = torch.linalg.cross(a, b, dim=1)
        y1 = y1 / (torch.sqrt((y1 ** 2).sum(1, keepdim=True)) + EPS)
        z1 = torch.linalg.cross(x1, y1, dim=1)
    elif nodeframe == "exact":
        deg = _scatter_sum(torch.ones(i.numel(), dtype=gdt), j, N)       # n_s - 1 (0 for a lone node)
        b = torch.where(deg.unsqueeze(1) > 0, -a / deg.clamp(min=1).unsqueeze(1), torch.zeros_like(a))
        x1 = (a - b) / (torch.sqrt(((a - b) ** 2).sum(1, keepdim=True)) + EPS)
        y1 = torch.zeros_like(a)
        z1 = torch.zeros_like(a)
    else:
        raise ValueError(nodeframe)
    nodeframe_t = torch.stack((x1, y1, z1), dim=-1)                       # [N,3(x),3(k)]
    pos_prjt = torch.sum(pos_frame.unsqueeze(-1) * nodeframe_t, dim=1)    # :834

    mask, pos_frame, dist = mask.to(dt), pos_frame.to(dt), dist.to(dt)
    coord_diff, coord_cross, coord_vertical = coord_diff.to(dt), coord_cross.to(dt), coord_vertical.to(dt)
    radial_emb, rbounds = radial_emb.to(dt), rbounds.to(dt)
    nodeframe_t, pos_prjt = nodeframe_t.to(dt), pos_prjt.to(dt)
    frame = torch.stack((coord_diff, coord_cross, coord_vertical), dim=-1)  # [E,3(x),3(k)] :773-780
    st["edge_mask"] = mask.squeeze(-1)
    st["pos_frame"] = pos_frame
    st["dist"] = dist
    st["coord_diff"] = coord_diff
    st["frame"] = frame
    st["radial_emb"] = radial_emb
    st["nodeframe"] = nodeframe_t
    st["pos_prjt"] = pos_prjt
    # ---- end of geometry block ------------------------------------------------------------------

    f = _linear(_silu(_linear(radial_emb, sd, p + "radial_lin.0")), sd, p + "radial_lin.2")  # :784
    f = rbounds.unsqueeze(-1) * f                                         # :786
    st["f"] = f

    # NeighborEmb (leftnet.py:81-89): PyG gathers x_j from edge_index[0], sums at edge_index[1]
    nb = _layer_norm(_linear(h, sd, p + "neighbor_emb.embedding"))
    s = z_emb + _scatter_sum(f * nb[i], j, N)                             # :789
    st["s0"] = s

    # CFConvS2V (leftnet.py:104-125)
    s1 = _silu(_layer_norm(_linear(s, sd, p + "s2v.lin1.0")))
    emb = f.unsqueeze(1) * coord_diff.unsqueeze(-1)                       # [E,3,H]
    NE1 = _scatter_sum(emb * s1[i].unsqueeze(1), j, N)                    # [N,3,H] :791
    st["NE1"] = NE1

    # edge scalarisation (leftnet.py:792-809)
    sc1 = torch.sum(NE1[i].unsqueeze(2) * frame.unsqueeze(-1), dim=1)     # [E,3(k),H]
    sc2 = torch.sum(NE1[j].unsqueeze(2) * frame.unsqueeze(-1), dim=1)
    if reflect_equiv:
        sc1 = torch.cat((sc1[:, :1], sc1[:, 1:2].abs(), sc1[:, 2:]), dim=1)
        sc2 = torch.cat((sc2[:, :1], sc2[:, 1:2].abs(), sc2[:, 2:]), dim=1)

    def lin3(x: Tensor) -> Tensor:  # x [E,H,3]
        def f(y: Tensor) -> Tensor:
            return _linear(_silu(_linear(y, sd, p + "lin3.0")), sd, p + "lin3.2")
        rows = max(1, (1 << 27) // max(1, x.shape[1] * (H // 4)))      # bound the [rows,H,H/4] intermediate (~1 GiB)
        if x.shape[0] <= rows:
            return f(x)
        return torch.cat([f(x[k:k + rows]) for k in range(0, x.shape[0], rows)], dim=0)

    sc1p = sc1.permute(0, 2, 1)
    sc2p = sc2.permute(0, 2, 1)
    scalar3 = (lin3(sc1p) + sc1p[:, :, 0].unsqueeze(2)).squeeze(-1)
    scalar4 = (lin3(sc2p) + sc2p[:, :, 0].unsqueeze(2)).squeeze(-1)
    edgeweight = torch.cat((scalar3, scalar4), dim=-1) * rbounds.unsqueeze(-1)
    edgeweight = torch.cat((edgeweight, f, radial_emb), dim=-1)           # [E, 3H+R]
    st["edgeweight0"] = edgeweight

    vec = torch.zeros(N, 3, H, dtype=dt)
    inv_sqrt_2 = 1 / math.sqrt(2.0)
    inv_sqrt_3 = 1 / math.sqrt(3.0)
    inv_sqrt_h = 1 / math.sqrt(H)
    cnt = _scatter_sum(torch.ones(i.numel(), dtype=dt), i, N).clamp(min=1)  # util_funcs.py:40-44

    for l in range(L):                                                    # :838
        s = s + _mlp(pos_prjt, sd, p + "pos_expansion", 2, last_layer_no_activation=True)  # :840-841 (legacy)

        # GCLMessage (leftnet.py:157-183)
        g = p + f"gcl_layers.{l}"
        xh = _layer_norm(s, sd[g + ".x_layernorm.weight"], sd[g + ".x_layernorm.bias"])
        m = _mlp(torch.cat([xh[i], xh[j], edgeweight], dim=1), sd, g + ".edge_mlp", 2)
        m = m * _mlp(m, sd, g + ".att_mlp", 1)                            # SiLU gate
        agg = _scatter_sum(m, i, N) / cnt.unsqueeze(1)                    # mean at edge_index[0]
        s = xh + _mlp(torch.cat([xh, agg], dim=1), sd, g + ".node_mlp", 2, last_layer_no_activation=True)
        edgeweight = edgeweight + _mlp(m, sd, g + ".edge_out_trans", 1)
        st[f"l{l}.s_gcl"] = s
        st[f"l{l}.edgeweight"] = edgeweight

        # EquiMessage (leftnet.py:244-284); PyG: *_j <- edge_index[0], *_i <- edge_index[1], sum at [1]
        q = p + f"message_layers.{l}"
        xq = _layer_norm(s, sd[q + ".x_layernorm.weight"], sd[q + ".x_layernorm.bias"])
        xq = _linear(_silu(_linear(xq, sd, q + ".x_proj.0")), sd, q + ".x_proj.2")
        rbfh = _linear(radial_emb, sd, q + ".rbf_proj")
        w = _linear(_silu(_linear(edgeweight, sd, q + ".dir_proj.0")), sd, q + ".dir_proj.2")
        rbfh = rbfh * w
        msg = (xq[i] + xq[j]) * rbfh
        x_m, xh2, xh3 = torch.split(msg, H, dim=-1)
        xh2 = xh2 * inv_sqrt_3
        vmsg = vec[i] * xh2.unsqueeze(1) + xh3.unsqueeze(1) * coord_diff.unsqueeze(2)
        if not reflect_equiv:
            vmsg = vmsg + x_m.unsqueeze(1) * coord_cross.unsqueeze(2)
        vmsg = vmsg * inv_sqrt_h
        dx = _scatter_sum(x_m, j, N)
        dvec = _scatter_sum(vmsg, j, N)
        st[f"l{l}.dx_msg"] = dx
        st[f"l{l}.dvec_msg"] = dvec
        s = (s + dx) * inv_sqrt_2                                         # :857-859
        vec = vec + dvec

        # EquiUpdate (leftnet.py:325-346)
        u = p + f"update_layers.{l}"
        vp = vec @ sd[u + ".vec_proj.weight"].t()
        vec1, vec2 = torch.split(vp, H, dim=-1)
        scal = torch.sum(vec1.unsqueeze(2) * nodeframe_t.unsqueeze(-1), dim=1)  # [N,3(k),H]
        if reflect_equiv:
            scal = torch.cat((scal[:, :1], scal[:, 1:2].abs(), scal[:, 2:]), dim=1)
        t3 = scal.permute(0, 2, 1)
        t3 = _silu(_linear(t3, sd, u + ".lin3.0"))
        t3 = _silu(_linear(t3, sd, u + ".lin3.2"))
        scalar = _linear(t3, sd, u + ".lin3.4").squeeze(-1)
        vec_dot = (vec1 * vec2).sum(dim=1) * inv_sqrt_h
        xv = torch.cat([s, scalar], dim=-1) @ sd[u + ".xvec_proj.0.weight"].t()
        xv = _silu(xv) @ sd[u + ".xvec_proj.2.weight"].t()
        xv1, xv2, xv3 = torch.split(xv, H, dim=-1)
        s = s + (xv1 + xv2 + vec_dot) * inv_sqrt_2                        # :861-864
        vec = vec + xv3.unsqueeze(1) * vec2
        st[f"l{l}.s"] = s
        st[f"l{l}.vec"] = vec

    # EquiOutput / GatedEquivariantBlock (leftnet.py:566-576), tail :878-891
    o = p + "out_pos.output_network.0"
    v1 = torch.norm(vec @ sd[o + ".vec1_proj.weight"].t(), dim=-2)      # :567 (torch.norm: zero subgradient at 0)
    v2 = vec @ sd[o + ".vec2_proj.weight"].t()                             # [N,3,1]
    xg = _linear(_silu(_linear(torch.cat([s, v1], dim=-1), sd, o + ".update_net.0")), sd, o + ".update_net.2")
    gate = xg[:, 1:2]
    dpos = (gate.unsqueeze(1) * v2).squeeze(-1)                           # [N,3]
    h_out = _linear(s, sd, p + "embedding_out")                           # :887
    st["dpos"] = dpos
    st["h_out"] = h_out
    return h_out, dpos


# --------------------------------------------------------------------------------------
# EGNNDynamics.forward (egnn_dynamics.py:63-168)
# --------------------------------------------------------------------------------------
def dynamics_forward(
    sd: Dict[str, Tensor],
    cfg: Dict,
    xh: List[Tensor],
    edge_index: Tensor,
    t: Tensor,
    conditions: Tensor,
    n_frag_switch: Tensor,
    combined_mask: Tensor,
    condition_nf: int,
    pos_dim: int = 3,
    condition_time: bool = True,
    encoder_alias: Optional[Sequence[int]] = None,
    nodeframe: str = "literal",
    stages: Optional[Dict[str, Tensor]] = None,
    direct_vel: bool = True,
    geom64: bool = False,
) -> List[Tensor]:
    """Returns the list of per-object [n_k, node_nf_k] tensors (vel || decoded h).

    `encoder_alias[k]` = index of the encoder/decoder object k uses (reference:
    `enforce_same_encoding`, _base.py:110-113).  `direct_vel=True` uses the model's
    `dpos` as the velocity; the reference forms `(pos + dpos) - pos`
    (egnn_dynamics.py:137, leftnet.py:882), identical in exact arithmetic.
    The NaN -> randn guard (egnn_dynamics.py:138-143) is not restated (RNG).
    """
    n_obj = len(xh)
    alias = list(encoder_alias) if encoder_alias is not None else list(range(n_obj))
    st = stages if stages is not None else {}
    pos = torch.cat([x[:, :pos_dim] for x in xh], dim=0)                  # :91-94
    h = torch.cat(
        [_mlp(xh[k][:, pos_dim:], sd, f"encoders.{alias[k]}", 2, last_layer_no_activation=True)
         for k in range(n_obj)], dim=0)                                   # :95-101
    condition_dim = 0
    if condition_time:
        if t.dim() == 1:
            h_time = torch.full_like(h[:, 0:1], float(t.item()))          # :108-110
        else:
            h_time = t[combined_mask]                                     # :112
        h = torch.cat([h, h_time.to(h.dtype)], dim=1)
        condition_dim += 1
    if condition_nf > 0:
        h = torch.cat([h, conditions[combined_mask].to(h.dtype)], dim=1)  # :116-119
        condition_dim += condition_nf
    st["h_in"] = h
    st["pos"] = pos

    subgraph_mask = get_subgraph_mask(edge_index, n_frag_switch)          # :121
    h_final, dpos = leftnet_forward(sd, cfg, h, pos, edge_index, subgraph_mask,
                                    nodeframe=nodeframe, stages=st, geom64=geom64)
    vel = dpos if direct_vel else (pos + dpos) - pos                      # :137
    h_final = h_final[:, :-condition_dim] if condition_dim else h_final   # :145

    # compute_frag_index (:177-182): rows of each object, objects in ascending id order
    counts = [int((n_frag_switch == k).sum()) for k in torch.unique(n_frag_switch).tolist()]
    frag_index = [0]
    for c in counts:
        frag_index.append(frag_index[-1] + c)
    out = []
    for k in range(n_obj):
        lo, hi = frag_index[k], frag_index[k + 1]
        v = vel[lo:hi]
        idx = combined_mask[lo:hi]
        if v.size(0):
            v = v - _scatter_mean(v, idx, int(idx.max()) + 1)[idx]        # :268-271
        hk = _mlp(h_final[lo:hi], sd, f"decoders.{alias[k]}", 2, last_layer_no_activation=True)
        out.append(torch.cat([v, hk], dim=-1))
    return out
