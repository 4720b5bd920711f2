"""Torch restatements of the node-side / init stages of the forward, in the library's internal node / edge order, unpadded
widths: TEST REFERENCES.  Each function restates exactly what the corresponding HIP forward kernel computes (reference lines
cited per function); tests/_stage_checks.py differentiates them with torch autograd on the training tape's stage inputs and
compares the result with the hand-written HIP adjoints (oard_train_stage_backward & co.), stage by stage, so that no stage sees
another stage's error.  The product path (oareactdiff_amd/training.py) does not use them."""
import ctypes as C
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.utils.checkpoint import checkpoint

from oareactdiff_amd import _capi
from oareactdiff_amd.training import _wgrad

INV_SQRT2, INV_SQRT3 = 1.0 / math.sqrt(2.0), 1.0 / math.sqrt(3.0)


def _seg_sum(x: Tensor, index: Tensor, n: int) -> Tensor:
    return torch.zeros((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device).index_add_(0, index, x)


def _ln(x: Tensor, w: Optional[Tensor] = None, b: Optional[Tensor] = None) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


# =====================================================================================================================
# Stage functions: torch restatements of the node-side / init HIP kernels, internal order, unpadded widths.
# `P` maps state-dict names to the module's parameters (leaves of the local autograd graphs).
# =====================================================================================================================
class Geometry:
    """Constants of one forward (no gradient flows into positions: `pos_grad=False`, en_diffusion.py trains eps-pred)."""

    def __init__(self, src: Tensor, tgt: Tensor, node_sample: Tensor, node_group: Tensor, n_samples: int, n_groups: int,
                 geo: Tensor, rbf: Tensor, pp0: Tensor, x1: Tensor):
        self.src, self.tgt = src, tgt                  # inner edges (target-sorted): source / target node
        self.node_sample, self.node_group = node_sample, node_group
        self.n_samples, self.n_groups = n_samples, n_groups
        self.env = geo[:, 1]                           # cosine envelope (leftnet.py:785)
        self.u = geo[:, 2:5]                           # coord_diff, masked (leftnet.py:693-705, 769)
        self.frame = torch.stack((geo[:, 2:5], geo[:, 5:8], geo[:, 8:11]), dim=-1)      # [A,3(x),3(k)]
        self.rbf = rbf                                 # [A,R] radial basis, masked (leftnet.py:781-782)
        self.pos_prjt = torch.stack((pp0, torch.zeros_like(pp0), torch.zeros_like(pp0)), dim=1)   # [N,3], exact frame
        self.x1 = x1                                   # [N,3]
        self.cross = geo[:, 5:8]                       # coord_cross, masked (leftnet.py:698-702, 770)
        self.reflect_equiv = True                      # model_config["reflect_equiv"]; False: signed scalarisation, x (x) coord_cross


def _lin3_rows(S: Tensor, w0: Tensor, b0: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """lin3 (leftnet.py:637-641, applied :798-805) on S [rows,3,H] -> [rows,H] (+ S[:,0])."""
    x = S.permute(0, 2, 1)                             # [rows,H,3]
    return (F.silu(x @ w0.t() + b0) @ w2.t() + b2).squeeze(-1) + S[:, 0]


def stage_init_head(P: Dict[str, Tensor], hin: Tensor, g: Geometry, H: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """k_node_embed, k_radial_lin, k_neighbor_v1, k_s2v_agg_v1, k_c0row (leftnet.py:744, 781-791).
    Returns (s0 [N,H], NE1 [N,3,H], f [A,H] = radial_lin(rbf) * envelope, constant row of the inter-object edges [3H+R])."""
    m = "model."
    N = hin.shape[0]
    z_emb = F.linear(hin, P[m + "embedding.weight"], P[m + "embedding.bias"])                              # :744
    nbe = _ln(F.linear(hin, P[m + "neighbor_emb.embedding.weight"], P[m + "neighbor_emb.embedding.bias"]))   # :82
    rl0w, rl0b, rl2w, rl2b = (P[m + "radial_lin.0.weight"], P[m + "radial_lin.0.bias"], P[m + "radial_lin.2.weight"],
                              P[m + "radial_lin.2.bias"])
    f = F.linear(F.silu(F.linear(g.rbf, rl0w, rl0b)), rl2w, rl2b) * g.env[:, None]                          # :784-786
    c0f = F.linear(F.silu(rl0b), rl2w, rl2b)           # radial_lin(0) * envelope(0): the f section of a masked edge
    c0s = (F.silu(P[m + "lin3.0.bias"]) @ P[m + "lin3.2.weight"].t() + P[m + "lin3.2.bias"]).reshape(())    # lin3(0) + 0
    c0 = torch.cat([c0s.expand(2 * H), c0f, torch.zeros(g.rbf.shape[1], dtype=f.dtype, device=f.device)])
    # NeighborEmb (:81-89): sum over ALL incoming edges; inter-object ones carry the constant f
    inter = _seg_sum(nbe, g.node_sample, g.n_samples)[g.node_sample] - _seg_sum(nbe, g.node_group, g.n_groups)[g.node_group]
    s0 = z_emb + _seg_sum(f * nbe[g.src], g.tgt, N) + c0f * inter
    s1 = F.silu(_ln(F.linear(s0, P[m + "s2v.lin1.0.weight"], P[m + "s2v.lin1.0.bias"])))                   # :116
    NE1 = _seg_sum((f * s1[g.src])[:, None, :] * g.u[:, :, None], g.tgt, N)                                 # [N,3,H] :117-125
    return s0, NE1, f, c0


def stage_scalarize(P: Dict[str, Tensor], NE1: Tensor, g: Geometry, H: int, chunk: int = 8192) -> Tensor:
    """k_scalarize (leftnet.py:792-806): [A, 2H] = (lin3(frame^T NE1[node]) + S_0) * envelope for node = source | target.
    torch restatement, used by the tests as the reference of the HIP backward `oard_scalarize_backward` (the product's
    backward never materialises the [rows, H, H/4] hidden layer this formulation needs)."""
    m = "model."
    l0w, l0b, l2w, l2b = P[m + "lin3.0.weight"], P[m + "lin3.0.bias"], P[m + "lin3.2.weight"], P[m + "lin3.2.bias"]
    parts = []
    A = g.src.numel()
    for a0 in range(0, A, chunk):                      # checkpointed chunks bound the [rows,H,H/4] intermediate
        sl = slice(a0, min(A, a0 + chunk))

        def piece(NE1_, l0w_, l0b_, l2w_, l2b_, sl=sl):
            fr = g.frame[sl]
            out = []
            for node in (g.src[sl], g.tgt[sl]):
                S = torch.einsum("axh,axk->akh", NE1_[node], fr)                                            # :792-793
                if g.reflect_equiv:
                    S = torch.cat((S[:, :1], S[:, 1:2].abs(), S[:, 2:]), dim=1)                             # :794-796
                out.append(_lin3_rows(S, l0w_, l0b_, l2w_, l2b_) * g.env[sl, None])
            return torch.cat(out, dim=1)
        parts.append(checkpoint(piece, NE1, l0w, l0b, l2w, l2b, use_reentrant=False) if A > chunk
                     else piece(NE1, l0w, l0b, l2w, l2b))
    return torch.cat(parts, dim=0) if parts else torch.zeros(0, 2 * H, dtype=NE1.dtype, device=NE1.device)


def stage_init(P: Dict[str, Tensor], hin: Tensor, g: Geometry, H: int) -> Tuple[Tensor, Tensor, Tensor]:
    """All init stages: (s0 [N,H], initial inner edge state [A,3H+R] (:806-809), constant inter-object row [3H+R])."""
    s0, NE1, f, c0 = stage_init_head(P, hin, g, H)
    return s0, torch.cat([stage_scalarize(P, NE1, g, H), f, g.rbf], dim=1), c0


def stage_node_pre(P: Dict[str, Tensor], l: int, s_in: Tensor, g: Geometry, H: int) -> Tuple[Tensor, Tensor, Tensor]:
    """k_node_pre_v1: s += pos_expansion(pos_prjt) (:840-841); xh = LN(s) (:158); node halves of edge_mlp.0 (:168)."""
    m = "model."
    pe = F.linear(F.silu(F.linear(g.pos_prjt, P[m + "pos_expansion.mlp.0.linear.weight"])),
                  P[m + "pos_expansion.mlp.1.linear.weight"])
    q = m + f"gcl_layers.{l}."
    xh = _ln(s_in + pe, P[q + "x_layernorm.weight"], P[q + "x_layernorm.bias"])
    w1 = P[q + "edge_mlp.mlp.0.linear.weight"]
    return xh, F.linear(xh, w1[:, :H], P[q + "edge_mlp.mlp.0.linear.bias"]), F.linear(xh, w1[:, H:2 * H])


def stage_gcl_node(P: Dict[str, Tensor], l: int, xh: Tensor, agg: Tensor, H: int) -> Tuple[Tensor, Tensor]:
    """k_gcl_node_v1: GCL node update (:172-183) and EquiMessage's node part x_proj (:245) -> (s_mid [N,H], xq [N,3H])."""
    m = "model."
    q = m + f"gcl_layers.{l}."
    hm = F.silu(F.linear(torch.cat([xh, agg], dim=1), P[q + "node_mlp.mlp.0.linear.weight"], P[q + "node_mlp.mlp.0.linear.bias"]))
    s = xh + F.linear(hm, P[q + "node_mlp.mlp.1.linear.weight"], P[q + "node_mlp.mlp.1.linear.bias"])
    e = m + f"message_layers.{l}."
    xq = F.linear(F.silu(F.linear(_ln(s, P[e + "x_layernorm.weight"], P[e + "x_layernorm.bias"]), P[e + "x_proj.0.weight"])),
                  P[e + "x_proj.2.weight"])
    return s, xq


def stage_equi_message(P: Dict[str, Tensor], l: int, s: Tensor, xq: Tensor, cd: Tensor, vec_in: Tensor, g: Geometry,
                       H: int) -> Tuple[Tensor, Tensor]:
    """Gather half of k_equi_node_v1: message formation and aggregation (:264-283, 857-859) -> (s_a, vec_a).
    torch restatement: the reference of the HIP adjoint `oard_equi_msg_backward` in the tests (the product's backward does
    not run this [A, 3H]-sized gather / scatter chain)."""
    N = s.shape[0]
    cr = F.linear(g.rbf, P[f"model.message_layers.{l}.rbf_proj.weight"])             # [A,3H]
    msg = (xq[g.src] + xq[g.tgt]) * (cd.reshape(cd.shape[0], 3 * H) * cr)
    x_m, a2, a3 = torch.split(msg, H, dim=-1)
    vmsg = vec_in[g.src] * (a2 * INV_SQRT3)[:, None, :] + a3[:, None, :] * g.u[:, :, None]
    if not g.reflect_equiv:
        vmsg = vmsg + x_m[:, None, :] * g.cross[:, :, None]                          # :268-272
    vmsg = vmsg * (1.0 / math.sqrt(H))
    return (s + _seg_sum(x_m, g.tgt, N)) * INV_SQRT2, vec_in + _seg_sum(vmsg, g.tgt, N)


class Lin3uFunction(torch.autograd.Function):
    """EquiUpdate's frame-scalar MLP (leftnet.py:304-310, 333) on [N, H] items through oard_lin3u_forward / _backward;
    the weight gradients are reduced by oard_wgrad.  `hip` = (dyn, cfg, layer, stream)."""

    @staticmethod
    def forward(ctx, sc, w0, b0, w2, b2, w4, b4, hip):
        dyn, cfg, layer, stream = hip
        x = sc.contiguous()
        out = torch.empty_like(x)
        packed = dyn._get_packed(cfg, stream)
        _capi.check(_capi.lib().oard_lin3u_forward(C.byref(cfg), packed.data_ptr(), layer, x.data_ptr(), x.numel(), out.data_ptr(),
                                                   stream), "oard_lin3u_forward")
        ctx.save_for_backward(x)
        ctx.hip = hip
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dyn, cfg, layer, stream = ctx.hip
        n, dev = x.numel(), x.device
        g = dout.contiguous()
        dx = torch.empty_like(x)
        xa, h1, dz1 = torch.empty(n, 4, device=dev), torch.empty(n, 48, device=dev), torch.empty(n, 48, device=dev)
        h2a, dz2 = torch.empty(n, 12, device=dev), torch.empty(n, 8, device=dev)
        packed = dyn._get_packed(cfg, stream)
        _capi.check(_capi.lib().oard_lin3u_backward(C.byref(cfg), packed.data_ptr(), layer, x.data_ptr(), g.data_ptr(), n,
                                                    dx.data_ptr(), xa.data_ptr(), h1.data_ptr(), dz1.data_ptr(), h2a.data_ptr(),
                                                    dz2.data_ptr(), stream), "oard_lin3u_backward")
        g0 = _wgrad(dz1, 48, 48, 48, 48, xa, 4, False, 2, 2, 2, n, False, dyn, stream)[0]           # [48, 2] = (d w0[:,0] | d b0)
        gw0 = torch.zeros(48, 3, device=dev)
        gw0[:, 0] = g0[:, 0]
        gw2, gb2 = _wgrad(dz2, 8, 8, 8, 8, h1, 48, False, 48, 48, 48, n, True, dyn, stream)
        g4 = _wgrad(h2a, 12, 9, 9, 9, xa, 4, False, 1, 1, 1, n, True, dyn, stream)[1]                # column sums of h2a
        return dx, gw0, g0[:, 1].contiguous(), gw2, gb2, g4[:8].view(1, 8), g4[8:9], None


def stage_equi_update(P: Dict[str, Tensor], l: int, s: Tensor, vec: Tensor, g: Geometry, H: int, hip=None) -> Tuple[Tensor, Tensor]:
    """Second half of k_equi_node_v1: EquiUpdate (:325-346, 861-864) on the aggregated state -> (s_out, vec_out).
    `hip` = (dyn, cfg, layer, stream): the frame-scalar MLP runs as the HIP op `Lin3uFunction` (product path); None: the
    plain torch formulation (tests: the reference of that op)."""
    u = f"model.update_layers.{l}."
    v1, v2 = torch.split(vec @ P[u + "vec_proj.weight"].t(), H, dim=-1)             # [N,3,H] each
    sc = (v1 * g.x1[:, :, None]).sum(dim=1)                                          # nodeframe = [x1, 0, 0]
    if hip is not None:
        scalar = Lin3uFunction.apply(sc, P[u + "lin3.0.weight"], P[u + "lin3.0.bias"], P[u + "lin3.2.weight"], P[u + "lin3.2.bias"],
                                     P[u + "lin3.4.weight"], P[u + "lin3.4.bias"], hip)
    else:
        t3 = torch.stack((sc, torch.zeros_like(sc), torch.zeros_like(sc)), dim=-1)   # [N,H,3]; |0| = 0 (:328-332)
        t3 = F.silu(F.linear(t3, P[u + "lin3.0.weight"], P[u + "lin3.0.bias"]))
        t3 = F.silu(F.linear(t3, P[u + "lin3.2.weight"], P[u + "lin3.2.bias"]))
        scalar = F.linear(t3, P[u + "lin3.4.weight"], P[u + "lin3.4.bias"]).squeeze(-1)
    vdot = (v1 * v2).sum(dim=1) * (1.0 / math.sqrt(H))
    xv = F.linear(F.silu(F.linear(torch.cat([s, scalar], dim=-1), P[u + "xvec_proj.0.weight"])), P[u + "xvec_proj.2.weight"])
    xa, xb, xc = torch.split(xv, H, dim=-1)
    return s + (xa + xb + vdot) * INV_SQRT2, vec + xc[:, None, :] * v2


def stage_node_mid(P: Dict[str, Tensor], l: int, xh: Tensor, agg: Tensor, cd: Tensor, vec_in: Tensor, g: Geometry,
                   H: int) -> Tuple[Tensor, Tensor]:
    """k_gcl_node_v1 + k_equi_node_v1 as one function (tests): cd [A,3,H] is dir_proj's output from the HIP edge kernel."""
    s, xq = stage_gcl_node(P, l, xh, agg, H)
    s, vec = stage_equi_message(P, l, s, xq, cd, vec_in, g, H)
    return stage_equi_update(P, l, s, vec, g, H)


def stage_out(P: Dict[str, Tensor], s: Tensor, vec: Tensor) -> Tuple[Tensor, Tensor]:
    """k_out_v1: GatedEquivariantBlock (:566-576) and the tail (:878-891) -> (dpos [N,3], h_out [N,C])."""
    o = "model.out_pos.output_network.0."
    v1 = torch.norm(vec @ P[o + "vec1_proj.weight"].t(), dim=-2)                       # :567, zero subgradient at vec = 0
    v2 = (vec @ P[o + "vec2_proj.weight"].t()).squeeze(-1)                            # [N,3]
    xg = F.linear(F.silu(F.linear(torch.cat([s, v1], dim=-1), P[o + "update_net.0.weight"], P[o + "update_net.0.bias"])),
                  P[o + "update_net.2.weight"], P[o + "update_net.2.bias"])
    return xg[:, 1:2] * v2, F.linear(s, P["model.embedding_out.weight"], P["model.embedding_out.bias"])




def stage_tail(P: Dict[str, Tensor], dec: list, s: Tensor, vec: Tensor, node_group: Tensor, n_groups: int, group_count: Tensor,
               obj_rows: list, node_row: Tensor, emb: int):
    """k_out_v1 + k_post: output block, velocity with the per-(sample, object) CoM removed, decoders (egnn_dynamics.py:137-160)
    -> one [n_k, 3 + d_k] tensor per object in the reference's row order."""
    dpos, hout = stage_out(P, s, vec)
    vel = dpos - (_seg_sum(dpos, node_group, n_groups) / group_count)[node_group]
    outs = []
    for k, rows in enumerate(obj_rows):
        hk = hout[rows, :emb]
        hk = F.linear(F.silu(F.linear(hk, P[dec[k] + "mlp.0.linear.weight"], P[dec[k] + "mlp.0.linear.bias"])),
                      P[dec[k] + "mlp.1.linear.weight"], P[dec[k] + "mlp.1.linear.bias"])
        o = torch.cat([vel[rows], hk], dim=1)
        outs.append(torch.zeros_like(o).index_copy(0, node_row[rows], o))      # internal -> row inside xh[k]
    return tuple(outs)


def stage_head(P: Dict[str, Tensor], enc: list, feats: list, node_ref: Tensor, hin_tail: Tensor) -> Tensor:
    """k_prep: per-object encoders (egnn_dynamics.py:95-104), rows brought into the internal order, time / condition columns appended."""
    hs = []
    for k, f in enumerate(feats):
        hs.append(F.linear(F.silu(F.linear(f, P[enc[k] + "mlp.0.linear.weight"], P[enc[k] + "mlp.0.linear.bias"])),
                           P[enc[k] + "mlp.1.linear.weight"], P[enc[k] + "mlp.1.linear.bias"]))
    return torch.cat([torch.cat(hs, dim=0)[node_ref], hin_tail], dim=1)
