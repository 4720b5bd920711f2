"""Backward pass of the denoising call.

`EGNNDynamics.forward` under autograd runs the HIP forward in training mode (`oard_forward_train`: same kernels, plus a
*tape* of per-layer edge state, pre-activations and node state) and returns tensors whose `grad_fn` is
`DynamicsFunction`.  Its backward is a hand-scheduled reverse sweep over the stages of the forward:

* the two per-layer **edge stages** (GCLMessage / EquiMessage edge parts, > 98 % of the FLOPs) are hand-written HIP
  kernels: `oard_gcl_backward_dx`, `oard_equi_backward_dx` (transposed-weight streams through LDS, fp32 MFMA) and
  `oard_wgrad` (the weight-gradient GEMM over edges) — csrc/oard_edge_bwd.h;
* the **node-side and init stages** (O(N) or O(A) element-wise / small-GEMM work, < 2 % of the FLOPs) are re-evaluated
  here as small torch-on-device functions from the taped stage inputs and differentiated by *local* autograd
  (`torch.autograd.grad` on that stage only).  Each stage function restates exactly what the corresponding HIP forward
  kernel computes (reference lines cited per function), in the library's internal node / edge order.

What the reference does instead: plain torch autograd through `LEFTNet.forward` (oa_reactdiff/model/leftnet.py:724-891)
from `DDPMModule.training_step` (oa_reactdiff/trainer/pl_trainer.py:327-347).  Gradients are produced for every
parameter of the state dict except the two modules the forward never touches (`model.distance_embedding`,
`model.last_layer`; the reference leaves their `.grad` None as well) and the RBF buffers.

There is no CPU fallback: the tape only exists on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.utils.checkpoint import checkpoint

from . import _capi

INV_SQRT2, INV_SQRT3 = 1.0 / math.sqrt(2.0), 1.0 / math.sqrt(3.0)


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


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
                S = torch.cat((S[:, :1], S[:, 1:2].abs(), S[:, 2:]), dim=1)                                 # :794-796
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
    vmsg = (vec_in[g.src] * (a2 * INV_SQRT3)[:, None, :] + a3[:, None, :] * g.u[:, :, None]) * (1.0 / math.sqrt(H))
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


# =====================================================================================================================
# The HIP side of one training-mode forward
# =====================================================================================================================
class TrainTopology:
    """parts == 1 topology plus the index tables the stage functions need (internal order, int64 on the device)."""

    def __init__(self, cfg, combined_mask: Tensor, n_frag_switch: Tensor, stream: int, edge_index: Optional[Tensor] = None):
        L = _capi.lib()
        dev = combined_mask.device
        cm = combined_mask.detach().to("cpu", torch.int64).contiguous()
        nfs = n_frag_switch.detach().to("cpu", torch.int64).contiguous()
        h = C.c_void_p()
        _capi.check(L.oard_topology_create_parts(C.byref(cfg), C.cast(cm.data_ptr(), C.POINTER(C.c_int64)),
                                                 C.cast(nfs.data_ptr(), C.POINTER(C.c_int64)), cm.numel(), 1, C.byref(h)),
                    "oard_topology_create_parts")
        self.handle, self._lib = h, L
        self.N, self.E = int(L.oard_topology_num_nodes(h)), int(L.oard_topology_num_edges(h))
        self.A, self.B = int(L.oard_topology_num_inner_edges(h)), int(L.oard_topology_num_samples(h))
        self.n_obj = cfg.n_obj
        self.max_sample_id = int(cm.max())
        self.obj_counts = [int((nfs == k).sum()) for k in range(cfg.n_obj)]

        def table(which: int, n: int) -> Tensor:
            t = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            _capi.check(L.oard_topology_export(h, which, t.data_ptr(), t.numel(), stream), "oard_topology_export")
            return t[:n].long()
        self.node_ref = table(_capi.TOPO_NODE_REF, self.N)
        self.node_obj = table(_capi.TOPO_NODE_OBJ, self.N)
        self.node_row = table(_capi.TOPO_NODE_ROW, self.N)
        self.node_sample = table(_capi.TOPO_NODE_SAMPLE, self.N)
        self.inner_src = table(_capi.TOPO_INNER_SRC, self.A)
        self.inner_tgt = table(_capi.TOPO_INNER_TGT, self.A)
        self.node_group = self.node_sample * cfg.n_obj + self.node_obj
        # rows of every object and the size of every (sample, object) group, built once here: a nonzero() in the backward sweep
        # would be a host sync right after the forward pass
        order = torch.argsort(self.node_obj, stable=True)
        starts = [0]
        for k in range(cfg.n_obj):
            starts.append(starts[-1] + self.obj_counts[k])
        self.obj_rows = [order[starts[k]:starts[k + 1]] for k in range(cfg.n_obj)]
        self.group_count = torch.zeros(self.B * cfg.n_obj, 1, device=dev).index_add_(
            0, self.node_group, torch.ones(self.N, 1, device=dev)).clamp(min=1)
        if edge_index is not None:       # the kernels assume the complete-per-sample graph in the reference's edge order: verify
            ei = edge_index.detach()
            if ei.dim() != 2 or ei.shape[0] != 2 or ei.dtype != torch.int64 or ei.device != dev:
                raise _capi.OardError("edge_index must be an int64 [2, E] tensor on the same device")
            ei = ei.contiguous()
            ok = torch.zeros(1, dtype=torch.int32, device=dev)
            _capi.check(L.oard_topology_check_edge_index(h, ei.data_ptr(), ei.shape[1], ok.data_ptr(), stream),
                        "oard_topology_check_edge_index")
            if int(ok.item()) != 1:
                raise _capi.OardError("edge_index is not get_edges_index(combined_mask, remove_self_edge=True)")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.oard_topology_destroy(self.handle)
        except Exception:
            pass


class Tape:
    """Views into the tape buffer of one oard_forward_train."""

    def __init__(self, cfg, topo: TrainTopology, buf: Tensor):
        self.cfg, self.topo, self.buf = cfg, topo, buf

    def get(self, which: int, layer: int = 0) -> Tensor:
        off, rows, ld = C.c_size_t(0), C.c_int64(0), C.c_int64(0)
        _capi.check(_capi.lib().oard_tape_entry(C.byref(self.cfg), self.topo.handle, which, layer, C.byref(off), C.byref(rows),
                                                C.byref(ld)), "oard_tape_entry")
        n = rows.value * ld.value
        return self.buf[off.value: off.value + 4 * n].view(torch.float32).view(rows.value, ld.value)


class TrainState:
    """Everything one forward leaves behind for its backward."""

    def __init__(self, cfg, topo: TrainTopology, tape: Tape, xh: List[Tensor], t: Tensor, t_scalar: bool,
                 cond: Optional[Tensor]):
        self.cfg, self.topo, self.tape = cfg, topo, tape
        self.xh, self.t, self.t_scalar, self.cond = xh, t, t_scalar, cond


def _wgrad(dY: Tensor, ncY: int, o_len: int, o_pad: int, MO: int, X: Tensor, ncX: int, x_silu: bool, i_len: int,
           i_pad: int, MI: int, rows: int, want_bias: bool, scratch_owner, stream: int) -> Tuple[Tensor, Optional[Tensor]]:
    """dW [MO,MI] (= nn.Linear weight shape) and db [MO] from row-major operands through oard_wgrad."""
    L = _capi.lib()
    need = L.oard_wgrad_scratch_bytes(ncY, ncX, rows)
    sc = getattr(scratch_owner, "_wgrad_scratch", None)
    if sc is None or sc.numel() < need or sc.device != dY.device:
        sc = torch.empty(need, dtype=torch.uint8, device=dY.device)
        scratch_owner._wgrad_scratch = sc
    dW = torch.empty(MO, MI, dtype=torch.float32, device=dY.device)
    db = torch.empty(MO, dtype=torch.float32, device=dY.device) if want_bias else None
    _capi.check(L.oard_wgrad(dY.data_ptr(), dY.stride(0), ncY, o_len, o_pad, MO, X.data_ptr(), X.stride(0), ncX,
                             1 if x_silu else 0, i_len, i_pad, MI, rows, dW.data_ptr(),
                             db.data_ptr() if db is not None else None, sc.data_ptr(), sc.numel(), stream), "oard_wgrad")
    return dW, db


def scalarize_backward(dyn, cfg, topo: TrainTopology, tape: Tape, NE1: Tensor, dew: Tensor, H: int, stream: int):
    """oard_scalarize_backward: adjoint of k_scalarize.  NE1 [N,3,H] contiguous, dew [E+1,WP] (gradient of the initial
    edge state, columns [0, 2H) of the inner rows are read) -> (d NE1 [N,3,H], {lin3 parameter name: gradient})."""
    L = _capi.lib()
    H4 = H // 4
    dNE1 = torch.empty_like(NE1)
    part = torch.empty(topo.N, 5 * H4 + 1, dtype=torch.float32, device=NE1.device)
    packed = dyn._get_packed(cfg, stream)
    _capi.check(L.oard_scalarize_backward(C.byref(cfg), topo.handle, packed.data_ptr(), tape.buf.data_ptr(), NE1.data_ptr(), H,
                                          dew.data_ptr(), dNE1.data_ptr(), part.data_ptr(), stream), "oard_scalarize_backward")
    tot = part.sum(dim=0)
    m = "model."
    return dNE1, {m + "lin3.0.weight": tot[: 3 * H4].view(H4, 3), m + "lin3.0.bias": tot[3 * H4: 4 * H4],
                  m + "lin3.2.weight": tot[4 * H4: 5 * H4].view(1, H4), m + "lin3.2.bias": tot[5 * H4:]}


def _local(fn: Callable, inputs: Sequence[Tensor], params: Dict[str, Tensor]):
    """Runs fn(*inputs) with the inputs and params as autograd leaves; returns (outputs, backward closure)."""
    with torch.enable_grad():
        ins = [x.detach().requires_grad_(True) for x in inputs]
        outs = fn(*ins)
    outs_t = outs if isinstance(outs, (tuple, list)) else (outs,)
    names = list(params)

    def backward(grad_outs: Sequence[Optional[Tensor]], grads: Dict[str, Tensor]) -> List[Optional[Tensor]]:
        pairs = [(o, g) for o, g in zip(outs_t, grad_outs) if g is not None and o.requires_grad]
        if not pairs:
            return [None] * len(ins)
        got = torch.autograd.grad([o for o, _ in pairs], ins + [params[n] for n in names], [g for _, g in pairs],
                                  allow_unused=True)
        for n, gp in zip(names, got[len(ins):]):
            if gp is not None:
                grads[n] = grads[n] + gp if n in grads else gp
        return list(got[: len(ins)])
    return [o.detach() for o in outs_t], backward


class _StageTimer:
    """OARD_TRAIN_PROFILE=1: device time per stage of the sweep (HIP events on the current stream), printed per call."""

    def __init__(self):
        import os
        self.on = bool(os.environ.get("OARD_TRAIN_PROFILE"))
        self.marks = []

    def mark(self, name: str):
        if self.on:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def report(self):
        if not self.on or len(self.marks) < 2:
            return
        torch.cuda.synchronize()
        tot: Dict[str, float] = {}
        for (n0, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
            tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
        print("backward sweep ms: " + "  ".join(f"{k} {v:.2f}" for k, v in tot.items()) + f"  | total {sum(tot.values()):.2f}")


def backward_sweep(dyn, st: TrainState, grad_outs: List[Optional[Tensor]], stream: int) -> Dict[str, Tensor]:
    """d(loss)/d(parameter) for every parameter the forward uses, given d(loss)/d(out[k])."""
    tm = _StageTimer()
    tm.mark("start")
    L = _capi.lib()
    cfg, topo, tape = st.cfg, st.topo, st.tape
    H, R, NL, Cc = dyn._dims
    W, HP, WP, D1P, RP = 3 * H + R, _pad16(H), _pad16(3 * H + R), _pad16(3 * H), _pad16(R)
    N, E, A = topo.N, topo.E, topo.A
    dev = st.xh[0].device
    P = dyn._param_dict()
    grads: Dict[str, Tensor] = {}
    emb = dyn.embed_dim
    n_obj = len(dyn.node_nfs)
    pd = dyn.pos_dim

    geo = tape.get(_capi.TAPE_GEO)[:A]
    g = Geometry(topo.inner_src, topo.inner_tgt, topo.node_sample, topo.node_group, topo.B, topo.B * n_obj, geo,
                 tape.get(_capi.TAPE_RBF)[:A, :R], tape.get(_capi.TAPE_PP0)[:, 0], tape.get(_capi.TAPE_X1))
    pbwd = dyn._get_packed_bwd(cfg, stream)
    enc = [dyn._module_prefix("encoders", k) for k in range(n_obj)]
    dec = [dyn._module_prefix("decoders", k) for k in range(n_obj)]

    def params_of(*prefixes: str) -> Dict[str, Tensor]:
        return {n: p for n, p in P.items() if n.startswith(prefixes)}

    # ---- output block + wrapper epilogue (k_out_v1, k_post; egnn_dynamics.py:137-160) ------------------------------
    s_L = tape.get(_capi.TAPE_S_IN, NL)[:, :H]
    vec_L = tape.get(_capi.TAPE_VEC_IN, NL).view(N, 3, HP)[:, :, :H]
    obj_rows = topo.obj_rows

    def tail(s, vec):
        dpos, hout = stage_out(P, s, vec)
        vel = dpos - (_seg_sum(dpos, topo.node_group, topo.B * n_obj) / topo.group_count)[topo.node_group]
        outs = []
        for k in range(n_obj):
            rows = obj_rows[k]
            hk = hout[rows, :emb]
            hk = F.linear(F.silu(F.linear(hk, P[dec[k] + "mlp.0.linear.weight"], P[dec[k] + "mlp.0.linear.bias"])),
                          P[dec[k] + "mlp.1.linear.weight"], P[dec[k] + "mlp.1.linear.bias"])
            o = torch.cat([vel[rows], hk], dim=1)
            outs.append(torch.zeros_like(o).index_copy(0, topo.node_row[rows], o))      # internal -> row inside xh[k]
        return tuple(outs)

    tail_params = params_of("model.out_pos.", "model.embedding_out.", *dec)
    _, bw = _local(tail, [s_L, vec_L], tail_params)
    ds, dvec = bw([None if go is None else go.to(torch.float32) for go in grad_outs], grads)
    tm.mark("tail")
    ds = torch.zeros(N, H, device=dev) if ds is None else ds
    dvec = torch.zeros(N, 3, H, device=dev) if dvec is None else dvec

    # gradient of the edge state leaving the current layer (padded, spare row included), updated in place by the kernels
    dew = torch.zeros(E + 1, WP, device=dev)
    dz3 = torch.empty(E + 1, WP, device=dev)
    mout, dz2, dz1 = (torch.empty(E + 1, HP, device=dev) for _ in range(3))
    da = torch.empty(E + 1, device=dev)
    dzd1 = torch.empty(A + 1, D1P, device=dev)
    dcd_p, dcr_p = torch.zeros(A + 1, 3, HP, device=dev), torch.zeros(A + 1, 3, HP, device=dev)   # pads / spare row stay zero
    dPQ = torch.empty(2, N, HP, device=dev)

    for l in reversed(range(NL)):
        q, e = f"model.gcl_layers.{l}.", f"model.message_layers.{l}."
        s_in = tape.get(_capi.TAPE_S_IN, l)[:, :H]
        vec_in = tape.get(_capi.TAPE_VEC_IN, l).view(N, 3, HP)[:, :, :H]
        agg = tape.get(_capi.TAPE_AGG, l)[:, :H]
        # node stages: k_node_pre_v1 graph first (its outputs feed both the edge kernel and the next node stage)
        (xh, _, _), bw_pre = _local(lambda s: stage_node_pre(P, l, s, g, H), [s_in],
                                    params_of("model.pos_expansion.", q + "x_layernorm.", q + "edge_mlp.mlp.0."))
        (_, xq), bw_gcl = _local(lambda a, b: stage_gcl_node(P, l, a, b, H), [xh, agg],
                                 params_of(q + "node_mlp.", e + "x_layernorm.", e + "x_proj."))
        s_a = tape.get(_capi.TAPE_S_A, l)[:, :H]
        vec_a = tape.get(_capi.TAPE_VEC_A, l).view(N, 3, HP)[:, :, :H]
        _, bw_upd = _local(lambda a, b: stage_equi_update(P, l, a, b, g, H, hip=(dyn, cfg, l, stream)), [s_a, vec_a],
                           params_of(f"model.update_layers.{l}."))
        tm.mark("node_fwd_recompute")
        gs_a, gvec_a = bw_upd([ds, dvec], grads)
        # ---- message formation + aggregation (HIP adjoint): -> d cd, d cr per edge, d xq, d vec entering the layer -------------
        gx = (gs_a * INV_SQRT2).contiguous()              # s_a = (s_mid + dx) / sqrt2
        cr = F.linear(g.rbf, P[e + "rbf_proj.weight"])    # [A,3H]
        dxq = torch.empty(N, 3 * H, device=dev)
        dvec = torch.empty(N, 3, H, device=dev)
        _capi.check(L.oard_equi_msg_backward(C.byref(cfg), topo.handle, tape.buf.data_ptr(), l, xq.contiguous().data_ptr(),
                                             cr.data_ptr(), gx.data_ptr(), gvec_a.contiguous().data_ptr(), dcd_p.data_ptr(),
                                             dcr_p.data_ptr(), dxq.data_ptr(), dvec.data_ptr(), stream), "oard_equi_msg_backward")
        if A > 0:
            grads[e + "rbf_proj.weight"] = _wgrad(dcr_p.view(A + 1, 3 * HP), 3 * HP, H, HP, 3 * H, tape.get(_capi.TAPE_RBF), RP, False,
                                                  R, R, R, A, False, dyn, stream)[0]
        else:
            grads[e + "rbf_proj.weight"] = torch.zeros(3 * H, R, device=dev)
        dxh, dagg = bw_gcl([gx, dxq], grads)
        tm.mark("node_mid_bwd")
        # ---- EquiMessage edge part (HIP): dcd -> dew[0:A], dir_proj gradients -------------------------------------------
        if A > 0:
            _capi.check(L.oard_equi_backward_dx(C.byref(cfg), topo.handle, pbwd.data_ptr(), l, tape.buf.data_ptr(),
                                                dcd_p.data_ptr(), dew.data_ptr(), dzd1.data_ptr(), stream),
                        "oard_equi_backward_dx")
            dcd2 = dcd_p.view(A + 1, 3 * HP)
            gw, gb = _wgrad(dcd2, 3 * HP, H, HP, 3 * H, tape.get(_capi.TAPE_ZD1, l), D1P, True, 3 * H, 3 * H, 3 * H, A,
                            True, dyn, stream)
            grads[e + "dir_proj.2.weight"], grads[e + "dir_proj.2.bias"] = gw, gb
            gw, gb = _wgrad(dzd1, D1P, 3 * H, 3 * H, 3 * H, tape.get(_capi.TAPE_EW, l + 1), WP, False, W, W, W, A, True,
                            dyn, stream)
            grads[e + "dir_proj.0.weight"], grads[e + "dir_proj.0.bias"] = gw, gb
        tm.mark("equi_edge_bwd+wgrad")
        # ---- GCLMessage edge part (HIP): dew (new state) + dagg -> dew (old state), dP, dQ, edge MLP gradients ------------
        dP = dQ = None
        if E > 0:
            dagg_p = torch.zeros(N, HP, device=dev)
            dagg_p[:, :H] = dagg
            _capi.check(L.oard_gcl_backward_dx(C.byref(cfg), topo.handle, pbwd.data_ptr(), l, tape.buf.data_ptr(),
                                               dagg_p.data_ptr(), dew.data_ptr(), dz3.data_ptr(), mout.data_ptr(),
                                               dz2.data_ptr(), da.data_ptr(), dz1.data_ptr(), stream), "oard_gcl_backward_dx")
            _capi.check(L.oard_edge_node_sums(C.byref(cfg), topo.handle, dz1.data_ptr(), dPQ[0].data_ptr(), dPQ[1].data_ptr(),
                                              stream), "oard_edge_node_sums")
            dP, dQ = dPQ[0, :, :H], dPQ[1, :, :H]
            rows3 = A if l == NL - 1 else E            # rows whose forward evaluated edge_out_trans
            if rows3 > 0:
                gw, gb = _wgrad(dz3, WP, W, W, W, mout, HP, False, H, H, H, rows3, True, dyn, stream)
            else:
                gw, gb = torch.zeros(W, H, device=dev), torch.zeros(W, device=dev)
            grads[q + "edge_out_trans.mlp.0.linear.weight"], grads[q + "edge_out_trans.mlp.0.linear.bias"] = gw, gb
            gw, gb = _wgrad(dz2, HP, H, H, H, tape.get(_capi.TAPE_Z1, l), HP, True, H, H, H, E, True, dyn, stream)
            grads[q + "edge_mlp.mlp.1.linear.weight"], grads[q + "edge_mlp.mlp.1.linear.bias"] = gw, gb
            # edge_mlp.0, edge-state columns: layer 0 sees the never-materialised constant row on inter-object edges
            rows1 = A if l == 0 else E
            g1c = (_wgrad(dz1, HP, H, H, H, tape.get(_capi.TAPE_EW, l), WP, False, W, W, W, rows1, False, dyn, stream)[0]
                   if rows1 > 0 else torch.zeros(H, W, device=dev))
            if l == 0 and E > A:
                g1c = g1c + torch.outer(dz1[A:E, :H].sum(dim=0), dyn._c0row(P, H, R))
            w1g = torch.zeros_like(P[q + "edge_mlp.mlp.0.linear.weight"])
            w1g[:, 2 * H:] = g1c
            grads[q + "edge_mlp.mlp.0.linear.weight"] = w1g      # node columns are added by bw_pre below
            m0 = F.silu(tape.get(_capi.TAPE_Z2, l)[:E, :H])
            grads[q + "att_mlp.mlp.0.linear.weight"] = (da[:E, None] * m0).sum(dim=0, keepdim=True)
            grads[q + "att_mlp.mlp.0.linear.bias"] = da[:E].sum().reshape(1)
        tm.mark("gcl_edge_bwd+wgrad")
        (ds,) = bw_pre([dxh, dP, dQ], grads)
        tm.mark("node_pre_bwd")
        ds = torch.zeros(N, H, device=dev) if ds is None else ds

    # ---- init stages + wrapper prologue (k_prep, k_node_embed, ..., k_scalarize; egnn_dynamics.py:91-119) ------------
    hin_tape = tape.get(_capi.TAPE_HIN)
    n_in = Cc
    feats = [x[:, pd:] for x in st.xh]

    def head():
        hs = []
        for k in range(n_obj):
            hk = F.linear(F.silu(F.linear(feats[k], P[enc[k] + "mlp.0.linear.weight"], P[enc[k] + "mlp.0.linear.bias"])),
                          P[enc[k] + "mlp.1.linear.weight"], P[enc[k] + "mlp.1.linear.bias"])
            hs.append(hk)
        h = torch.cat(hs, dim=0)[topo.node_ref]                  # reference (object-major) rows -> internal order
        return torch.cat([h, hin_tape[:, emb:n_in]], dim=1)      # time / condition columns are constants

    def init():
        return stage_init_head(P, head(), g, H)
    init_params = params_of("model.embedding.", "model.neighbor_emb.", "model.s2v.", "model.radial_lin.", "model.lin3.", *enc)
    (_, NE1, _, _), bw_init = _local(init, [], init_params)
    dNE1 = df = None
    if A > 0:
        # edge scalarisation + lin3 (k_scalarize): HIP adjoint -> d NE1 and the lin3 gradients
        dNE1, gl3 = scalarize_backward(dyn, cfg, topo, tape, NE1.contiguous(), dew, H, stream)
        for n_, g_ in gl3.items():
            grads[n_] = grads[n_] + g_ if n_ in grads else g_
        df = dew[:A, 2 * H:3 * H]
    dc0 = dew[A:E, :W].sum(dim=0) if E > A else None
    bw_init([ds, dNE1, df, dc0], grads)
    tm.mark("init_bwd")
    tm.report()
    return grads


class DynamicsFunction(torch.autograd.Function):
    """outs = EGNNDynamics(xh, ...) with a HIP forward (training mode) and the sweep above as backward.
    Inputs after the fixed arguments: the xh tensors, then every parameter (so autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, dyn, run_forward, n_obj, *tensors):
        outs, state = run_forward()
        ctx.dyn, ctx.state, ctx.n_obj = dyn, state, n_obj
        ctx.names = dyn._param_names()
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        dyn, st = ctx.dyn, ctx.state
        dev = st.xh[0].device
        with torch.cuda.device(dev), torch.no_grad():
            stream = torch.cuda.current_stream(dev).cuda_stream
            grads = backward_sweep(dyn, st, [g.contiguous() if g is not None else None for g in grad_outs], stream)
        ctx.state = None
        out: List[Optional[Tensor]] = [None, None, None] + [None] * ctx.n_obj       # no gradient w.r.t. the noised inputs
        for n in ctx.names:
            out.append(grads.get(n))
        return tuple(out)
