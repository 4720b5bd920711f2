"""The torch stage functions of tests/_stage_refs.py (the references the hand-written HIP adjoints of the node-side stages are
checked against on the GPU) composed into a whole forward on the CPU, in float64, against the oracle: every restatement and the internal
(sample-major node, target-sorted inner edge) layout conventions are checked without a GPU.  The two edge stages, which are
HIP kernels in the product, are stood in for by a few lines of torch here (on the GPU they are tested in
tests/test_grad_stages.py against the same stage functions)."""
import math

import pytest
import torch
import torch.nn.functional as F

import leftnet_oracle as oracle
from _cases import Case, rel
import _stage_refs as training


def _internal_layout(c, st):
    """Index tables of the library's internal order built independently (numpy-style) from combined_mask / n_frag_switch."""
    cm, nfs, ei = c.combined_mask, c.n_frag_switch, c.edge_index
    n_obj = c.n_obj
    order = torch.argsort(cm * n_obj + nfs, stable=True)                  # internal node -> reference row
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel())
    src, tgt = inv[ei[0]], inv[ei[1]]                                        # reference edges in internal node ids
    group = (cm * n_obj + nfs)[order]
    inner = (group[src] == group[tgt]).nonzero(as_tuple=True)[0]
    key = tgt[inner] * order.numel() + src[inner]
    inner = inner[torch.argsort(key)]                                        # inner edges sorted by (target, source)
    samples = torch.unique(cm)
    dense = torch.searchsorted(samples, cm)[order]
    return order, src, tgt, inner, dense, dense * n_obj + nfs[order], int(samples.numel())


@pytest.mark.parametrize("name", ["g2s_prod_b1_n5", "g3_cutoff_ragged", "g1_wrapper_small"])
def test_stage_functions_compose_to_the_oracle_forward(name):
    c = Case(name)
    cfg, H, R, NL = c.cfg, c.cfg["hidden_channels"], c.cfg["num_radial"], c.cfg["num_layers"]
    W = 3 * H + R
    sd = c.state_dict(torch.float64)
    st = {}
    ref = oracle.dynamics_forward(sd, cfg, [x.double() for x in c.xh], c.edge_index, c.t.double(), c.conditions.double(),
                                  c.n_frag_switch, c.combined_mask, c.cnf, nodeframe="exact", stages=st)
    order, src, tgt, inner, node_sample, node_group, B = _internal_layout(c, st)
    N, A = order.numel(), inner.numel()
    dist = st["dist"][inner]
    env = 0.5 * (torch.cos(dist * math.pi / float(cfg["cutoff"])) + 1.0)
    fr = st["frame"][inner]                                                   # [A, 3(x), 3(k)], masked
    geo = torch.cat([dist[:, None], env[:, None], fr[:, :, 0], fr[:, :, 1], fr[:, :, 2], st["edge_mask"][inner][:, None]], dim=1)
    g = training.Geometry(src[inner], tgt[inner], node_sample, node_group, B, B * c.n_obj, geo, st["radial_emb"][inner],
                          st["pos_prjt"][order, 0], st["nodeframe"][order][:, :, 0])
    P = sd
    s, ew_inner, c0 = training.stage_init(P, st["h_in"][order], g, H)
    assert rel(s, st["s0"][order]) < 1e-10
    ew = c0.expand(src.numel(), W).clone()
    ew[inner] = ew_inner
    assert rel(ew, st["edgeweight0"]) < 1e-10                                 # masked (inter-object) edges carry the constant row
    vec = torch.zeros(N, 3, H, dtype=torch.float64)
    deg = torch.zeros(N, dtype=torch.float64).index_add_(0, src, torch.ones(src.numel(), dtype=torch.float64)).clamp(min=1)
    for l in range(NL):
        q, e = f"model.gcl_layers.{l}.", f"model.message_layers.{l}."
        xh, Pn, Qn = training.stage_node_pre(P, l, s, g, H)
        # GCLMessage edge part (HIP kernel k_gcl_edge_v1 in the product)
        z1 = Pn[src] + Qn[tgt] + F.linear(ew, P[q + "edge_mlp.mlp.0.linear.weight"][:, 2 * H:])
        m0 = F.silu(F.linear(F.silu(z1), P[q + "edge_mlp.mlp.1.linear.weight"], P[q + "edge_mlp.mlp.1.linear.bias"]))
        m = m0 * F.silu(F.linear(m0, P[q + "att_mlp.mlp.0.linear.weight"], P[q + "att_mlp.mlp.0.linear.bias"]))
        agg = torch.zeros(N, H, dtype=torch.float64).index_add_(0, src, m) / deg[:, None]
        ew = ew + F.silu(F.linear(m, P[q + "edge_out_trans.mlp.0.linear.weight"], P[q + "edge_out_trans.mlp.0.linear.bias"]))
        s_mid, xq = training.stage_gcl_node(P, l, xh, agg, H)
        assert rel(s_mid, st[f"l{l}.s_gcl"][order]) < 1e-10
        # EquiMessage edge part (k_equi_edge_v1): dir_proj on the inner rows
        cd = F.linear(F.silu(F.linear(ew[inner], P[e + "dir_proj.0.weight"], P[e + "dir_proj.0.bias"])), P[e + "dir_proj.2.weight"],
                      P[e + "dir_proj.2.bias"]).view(A, 3, H)
        vec_prev = vec
        s_a, vec_a = training.stage_equi_message(P, l, s_mid, xq, cd, vec, g, H)
        s, vec = training.stage_equi_update(P, l, s_a, vec_a, g, H)
        assert rel(s, st[f"l{l}.s"][order]) < 1e-10 and rel(vec, st[f"l{l}.vec"][order]) < 1e-10
        # the fused restatement used by the stage tests is the same function
        s2, vec2 = training.stage_node_mid(P, l, xh, agg, cd, vec_prev, g, H)
        assert torch.equal(s2, s) and torch.equal(vec2, vec)
    dpos, hout = training.stage_out(P, s, vec)
    assert rel(dpos, st["dpos"][order]) < 1e-10 and rel(hout, st["h_out"][order]) < 1e-10
    # wrapper epilogue: per-(sample, object) centre-of-mass removal (egnn_dynamics.py:147-160) -> the oracle's outputs, reference order
    ng = B * c.n_obj
    mean = torch.zeros(ng, 3, dtype=torch.float64).index_add_(0, node_group, dpos) / \
        torch.zeros(ng, dtype=torch.float64).index_add_(0, node_group, torch.ones(N, dtype=torch.float64)).clamp(min=1)[:, None]
    vel = torch.empty_like(dpos)
    vel[order] = dpos - mean[node_group]                                      # back to the reference's (object-major) rows
    rv = torch.cat([o[:, :3] for o in ref], dim=0)
    assert rel(vel, rv) < 1e-10
