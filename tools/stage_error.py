"""Intrinsic error of every forward stage of the HIP path, beside plain torch float32 (round 6: where does the node state's error grow?).
A training-mode forward tapes every stage boundary; each stage is re-evaluated by its torch restatement (tests/_stage_refs.py) FROM THE
TAPED INPUTS in float64 (the truth for those inputs) and in float32 (what plain torch achieves): a stage whose HIP column is well above
its torch column is a summation-order / arithmetic problem of that kernel; equal columns mean the error is the conditioning of the stage.
usage (GPU box): python tools/stage_error.py [atoms per object] [B] [debug options k=v ...]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import _stage_refs as refs  # noqa: E402
from _stage_checks import tape_rows  # noqa: E402
from oareactdiff_amd import _capi, training  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402
from oareactdiff_amd.synthetic import make_inputs, make_topology  # noqa: E402


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    L = _capi.lib()
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        assert L.oard_debug_option(k.encode(), int(v)) == 0, k
    dev = torch.device("cuda:0")
    cfg_d = dict(PRODUCTION_LEFTNET_CONFIG)
    sd = synthetic_state_dict(state_spec(cfg_d, [9, 9, 9], 1), cfg_d, seed=42)
    dyn = EGNNDynamics(model_config=dict(cfg_d), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    cm, nfs, ei, masks = make_topology(B, nf)
    xh = make_inputs(B, nf, masks, 5, "cpu")
    t, cond = torch.full((B, 1), 0.6), torch.zeros(B, 1)
    keep = {}
    orig = training.DynamicsFunction.forward

    def spy(ctx, dyn_, run_forward, n_obj, *tensors):
        o = orig(ctx, dyn_, run_forward, n_obj, *tensors)
        keep["state"] = ctx.state
        return o
    training.DynamicsFunction.forward = staticmethod(spy)
    try:
        out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))      # autograd on: training-mode forward
    finally:
        training.DynamicsFunction.forward = orig
    st = keep["state"]
    tape, topo, cfg = st.tape, st.topo, st.cfg
    H, R, NL, Cc = dyn._dims
    HP, WP = training._pad16(H), training._pad16(3 * H + R)
    W = 3 * H + R
    N, E, A = topo.N, topo.E, topo.A
    P = {k: v.detach() for k, v in dyn._param_dict().items()}
    P64 = {k: v.double() for k, v in P.items()}
    geo = tape.get(_capi.TAPE_GEO)[:A]
    gargs = (topo.inner_src, topo.inner_tgt, topo.node_sample, topo.node_group, topo.B, topo.B * 3)
    rbf_t, pp0_t, x1_t = tape.get(_capi.TAPE_RBF)[:A, :R], tape.get(_capi.TAPE_PP0)[:, 0], tape.get(_capi.TAPE_X1)
    g = refs.Geometry(*gargs, geo, rbf_t, pp0_t, x1_t)
    g64 = refs.Geometry(*gargs, geo.double(), rbf_t.double(), pp0_t.double(), x1_t.double())
    g.reflect_equiv = g64.reflect_equiv = True
    stream = torch.cuda.current_stream(dev).cuda_stream
    rs = tape_rows(topo, L, dev, stream)
    src, tgt = rs["src"], rs["tgt"]
    print(f"N {N} E {E} A {A}; columns: HIP vs float64 | torch float32 vs float64 (both on the HIP path's taped stage inputs)")

    def row(tag, hip, f32, f64):
        print(f"   {tag:34s} {rel(hip, f64):.2e} | {rel(f32, f64):.2e}")
    with torch.no_grad():
        for l in range(NL):
            s_in = tape.get(_capi.TAPE_S_IN, l)[:, :H]
            vec_in = tape.get(_capi.TAPE_VEC_IN, l).view(N, 3, HP)[:, :, :H]
            ew = tape.get(_capi.TAPE_EW, l)[:E, :W]
            agg = tape.get(_capi.TAPE_AGG, l)[:, :H]
            s_mid = tape.get(_capi.TAPE_S_MID, l)[:, :H]
            s_a = tape.get(_capi.TAPE_S_A, l)[:, :H]
            vec_a = tape.get(_capi.TAPE_VEC_A, l).view(N, 3, HP)[:, :, :H]
            cd = tape.get(_capi.TAPE_CD, l)[:A].view(A, 3, HP)[:, :, :H]
            z1 = tape.get(_capi.TAPE_Z1, l)[:E, :H]
            z2 = tape.get(_capi.TAPE_Z2, l)[:E, :H]
            att = tape.get(_capi.TAPE_ATT, l)[:E, 0]
            s_out = tape.get(_capi.TAPE_S_IN, l + 1)[:, :H]
            vec_out = tape.get(_capi.TAPE_VEC_IN, l + 1).view(N, 3, HP)[:, :, :H]
            ew_out = tape.get(_capi.TAPE_EW, l + 1)[:E, :W]
            q = f"model.gcl_layers.{l}."
            res = {}
            for dt, Pd, gd in ((torch.float32, P, g), (torch.float64, P64, g64)):
                c = lambda x: x.to(dt)          # noqa: E731
                xh_, Pn, Qn = refs.stage_node_pre(Pd, l, c(s_in), gd, H)
                w1 = Pd[q + "edge_mlp.mlp.0.linear.weight"]
                z1_ = Pn[src] + Qn[tgt] + F.linear(c(ew), w1[:, 2 * H:])
                z2_ = F.linear(F.silu(c(z1)), Pd[q + "edge_mlp.mlp.1.linear.weight"], Pd[q + "edge_mlp.mlp.1.linear.bias"])
                m0 = F.silu(c(z2))
                att_ = F.linear(m0, Pd[q + "att_mlp.mlp.0.linear.weight"], Pd[q + "att_mlp.mlp.0.linear.bias"]).squeeze(-1)
                m_ = m0 * F.silu(c(att))[:, None]
                deg = torch.zeros(N, dtype=dt, device=dev).index_add_(0, src, torch.ones(E, dtype=dt, device=dev)).clamp(min=1)
                agg_ = torch.zeros(N, H, dtype=dt, device=dev).index_add_(0, src, m_) / deg[:, None]
                ew_ = c(ew) + F.silu(F.linear(m_, Pd[q + "edge_out_trans.mlp.0.linear.weight"], Pd[q + "edge_out_trans.mlp.0.linear.bias"]))
                s_mid_, xq_ = refs.stage_gcl_node(Pd, l, xh_, c(agg), H)
                s_a_, vec_a_ = refs.stage_equi_message(Pd, l, c(s_mid), refs.stage_gcl_node(Pd, l, xh_, c(agg), H)[1], c(cd), c(vec_in), gd, H)
                s_out_, vec_out_ = refs.stage_equi_update(Pd, l, c(s_a), c(vec_a), gd, H)
                res[dt] = dict(z1=z1_, z2=z2_, att=att_, agg=agg_, ew=ew_, s_mid=s_mid_, s_a=s_a_, vec_a=vec_a_, s_out=s_out_, vec_out=vec_out_)
            a, b = res[torch.float32], res[torch.float64]
            print(f"layer {l}")
            r1 = slice(0, A) if l == 0 else slice(0, E)      # layer 0: the inter-object rows' constant initial state is never materialised
            row("edge S1: z1 (xh, P, Q, ew taped in)", z1[r1], a["z1"][r1], b["z1"][r1])
            row("edge S2: z2 (z1 in)", z2, a["z2"], b["z2"])
            row("edge gate: att (z2 in)", att, a["att"], b["att"])
            row("mean message: agg (z2, att in)", agg, a["agg"], b["agg"])
            inner = slice(0, A) if l in (0, NL - 1) else slice(0, E)
            row("edge S3: ew_out (z2, att, ew in)", ew_out[inner], a["ew"][inner], b["ew"][inner])
            row("GCL node: s_mid (s_in, agg in)", s_mid, a["s_mid"], b["s_mid"])
            row("Equi gather: s_a (s_mid, cd in)", s_a, a["s_a"], b["s_a"])
            row("Equi gather: vec_a", vec_a, a["vec_a"], b["vec_a"])
            row("EquiUpdate: s_out (s_a, vec_a in)", s_out, a["s_out"], b["s_out"])
            row("EquiUpdate: vec_out", vec_out, a["vec_out"], b["vec_out"])
        s_f = tape.get(_capi.TAPE_S_IN, NL)[:, :H]
        vec_f = tape.get(_capi.TAPE_VEC_IN, NL).view(N, 3, HP)[:, :, :H]
        d32, h32 = refs.stage_out(P, s_f, vec_f)
        d64, h64 = refs.stage_out(P64, s_f.double(), vec_f.double())
        print(f"output block on the taped (s, vec): torch float32 dpos {rel(d32, d64):.2e} h_out {rel(h32, h64):.2e}  (max|dpos| {float(d64.abs().max()):.2e})")


if __name__ == "__main__":
    main()
