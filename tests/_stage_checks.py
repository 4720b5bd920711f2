"""Teacher-forced checks of the training path, shared by tests/test_grad_stages.py (asserts) and tools/grad_debug.py
(prints).  For one recorded training step (tests/golden/g9_grad_*.npz):
  1. every torch stage function of oareactdiff_amd/training.py, re-evaluated from the TAPED inputs of the HIP forward, against
     the TAPED outputs of the HIP kernels it restates (the local-autograd stages differentiate exactly these functions);
  2. the HIP backward kernels of the two edge stages (oard_gcl_backward_dx, oard_edge_node_sums, oard_equi_backward_dx,
     oard_wgrad) against torch autograd on a torch restatement of the same stage, fed with the tape's inputs and random
     cotangents - each kernel in isolation, nothing upstream can mask an error;
  3. the whole step against the reference's float64 gradients."""
import ctypes as C

import torch
import torch.nn.functional as F

from _grad_cases import CNF, NODE_NFS, GradCase
from oareactdiff_amd import _capi, training
from oareactdiff_amd.dynamics import EGNNDynamics


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))


def run(name, log=print):
    out = {}

    def note(key, *vals):
        out[key] = [float(v) for v in vals]
        log(f"{key:34s} " + "  ".join(f"{float(v):.2e}" for v in vals))
    c = GradCase(name)
    dev = torch.device("cuda:0")
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=NODE_NFS, edge_nf=0,
                       condition_nf=CNF, device=dev)
    dyn.load_state_dict(c.state_dict(), strict=True)
    keep = {}
    orig = training.DynamicsFunction.forward

    def spy(ctx, dyn_, run_forward, n_obj, *tensors):
        out = orig(ctx, dyn_, run_forward, n_obj, *tensors)
        keep["state"] = ctx.state
        return out
    training.DynamicsFunction.forward = staticmethod(spy)
    loss = c.loss(dyn, torch.float32, dev)
    st = keep["state"]
    tape, topo, cfg = st.tape, st.topo, st.cfg
    H, R, NL, Cc = dyn._dims
    HP, WP, D1P = training._pad16(H), training._pad16(3 * H + R), training._pad16(3 * H)
    W = 3 * H + R
    N, E, A = topo.N, topo.E, topo.A
    P = dyn._param_dict()
    log(f"{name}: N {N} E {E} A {A} loss {float(loss.detach()):.8f} ref64 {float(c.z['f64_loss']):.8f}")
    geo = tape.get(_capi.TAPE_GEO)[:A]
    g = training.Geometry(topo.inner_src, topo.inner_tgt, topo.node_sample, topo.node_group, topo.B, topo.B * 3, geo,
                          tape.get(_capi.TAPE_RBF)[:A, :R], tape.get(_capi.TAPE_PP0)[:, 0], tape.get(_capi.TAPE_X1))
    stream = torch.cuda.current_stream(dev).cuda_stream
    L = _capi.lib()
    with torch.no_grad():
        # ---- 1. forward consistency of the stage functions -------------------------------------------------------
        hin = tape.get(_capi.TAPE_HIN)[:, :Cc]
        s0, ew0, c0 = training.stage_init(P, hin, g, H)
        note("fwd init: s0, ew0", rel(s0, tape.get(_capi.TAPE_S_IN, 0)[:, :H]), rel(ew0, tape.get(_capi.TAPE_EW, 0)[:A, :W]))
        for l in range(NL):
            s_in = tape.get(_capi.TAPE_S_IN, l)[:, :H]
            vec_in = tape.get(_capi.TAPE_VEC_IN, l).view(N, 3, HP)[:, :, :H]
            xh, Pn, Qn = training.stage_node_pre(P, l, s_in, g, H)
            agg = tape.get(_capi.TAPE_AGG, l)[:, :H]
            cd = tape.get(_capi.TAPE_CD, l)[:A].view(A, 3, HP)[:, :, :H]
            s_out, vec_out = training.stage_node_mid(P, l, xh, agg, cd, vec_in, g, H)
            note(f"fwd layer {l}: s_out, vec_out", rel(s_out, tape.get(_capi.TAPE_S_IN, l + 1)[:, :H]),
                 rel(vec_out, tape.get(_capi.TAPE_VEC_IN, l + 1).view(N, 3, HP)[:, :, :H]))
    # ---- 2a. edge scalarisation + lin3 (k_scalarize): HIP adjoint vs torch autograd, random cotangent ------------------
    if A > 0:
        with torch.no_grad():
            _, NE1, _, _ = training.stage_init_head(P, hin, g, H)
        gen0 = torch.Generator(device="cpu").manual_seed(11)
        Gs = torch.randn(A, 2 * H, generator=gen0).to(dev)
        l3n = ["model.lin3.0.weight", "model.lin3.0.bias", "model.lin3.2.weight", "model.lin3.2.bias"]
        NE1t = NE1.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            sc = training.stage_scalarize(P, NE1t, g, H)
        gs = torch.autograd.grad([sc], [NE1t] + [P[n] for n in l3n], [Gs])
        dews = torch.zeros(E + 1, WP, device=dev)
        dews[:A, :2 * H] = Gs
        dNE1, gl3 = training.scalarize_backward(dyn, cfg, topo, tape, NE1.contiguous(), dews, H, stream)
        # d|S_1| carries sign(S_1).  Where S_1 = <NE1, coord_cross> is zero in exact arithmetic (three-atom objects are
        # coplanar with their centre of mass; (anti)parallel pairs when the cutoff bites) its computed value is rounding
        # noise and the sign is arbitrary - in the reference too.  Nodes touching such an item are left out.
        with torch.no_grad():
            cvec = g.frame[:, :, 1]
            s1_src, s1_tgt = torch.einsum("axh,ax->ah", NE1[g.src], cvec), torch.einsum("axh,ax->ah", NE1[g.tgt], cvec)
            thr = 1e-5 * max(float(s1_src.abs().max()), float(s1_tgt.abs().max()))
            noisy = torch.zeros(N, device=dev)
            noisy.index_add_(0, g.src, (s1_src.abs() <= thr).any(dim=1).float())
            noisy.index_add_(0, g.tgt, (s1_tgt.abs() <= thr).any(dim=1).float())
        ok = noisy == 0
        log(f"scalarize check: {int(ok.sum())} of {N} nodes have no sign-noisy item")
        note("bwd scalarize: dNE1, lin3 w0 b0 w2 b2", (rel(dNE1[ok], gs[0][ok]) if bool(ok.any()) else 0.0), *[rel(gl3[n].reshape(gs[1 + i].shape), gs[1 + i]) for i, n in enumerate(l3n)])
    # ---- 2. edge backward kernels, teacher-forced ----------------------------------------------------------------------
    rs = tape_rows(topo, L, dev, stream)
    pbwd = dyn._get_packed_bwd(cfg, stream)
    gen = torch.Generator(device="cpu").manual_seed(5)
    for l in range(NL):
        q, e = f"model.gcl_layers.{l}.", f"model.message_layers.{l}."
        last = l == NL - 1
        with torch.no_grad():
            s_in = tape.get(_capi.TAPE_S_IN, l)[:, :H]
            xh, Pn, Qn = training.stage_node_pre(P, l, s_in, g, H)
        c0row = dyn._c0row(P, H, R).detach()
        ew_l = tape.get(_capi.TAPE_EW, l)[:E, :W].clone()
        if l == 0:
            ew_l[A:] = c0row
        w1, b1 = P[q + "edge_mlp.mlp.0.linear.weight"], P[q + "edge_mlp.mlp.0.linear.bias"]
        names = [q + "edge_mlp.mlp.0.linear.weight", q + "edge_mlp.mlp.1.linear.weight", q + "edge_mlp.mlp.1.linear.bias",
                 q + "edge_out_trans.mlp.0.linear.weight", q + "edge_out_trans.mlp.0.linear.bias",
                 q + "att_mlp.mlp.0.linear.weight", q + "att_mlp.mlp.0.linear.bias"]
        ew_t = ew_l.detach().requires_grad_(True)
        Pt, Qt = Pn.detach().requires_grad_(True), Qn.detach().requires_grad_(True)
        with torch.enable_grad():
            z1 = Pt[rs["src"]] + Qt[rs["tgt"]] + F.linear(ew_t, w1[:, 2 * H:])
            m0 = F.silu(F.linear(F.silu(z1), P[q + "edge_mlp.mlp.1.linear.weight"], P[q + "edge_mlp.mlp.1.linear.bias"]))
            m = m0 * F.silu(F.linear(m0, P[q + "att_mlp.mlp.0.linear.weight"], P[q + "att_mlp.mlp.0.linear.bias"]))
            ew_new = ew_t + F.silu(F.linear(m, P[q + "edge_out_trans.mlp.0.linear.weight"], P[q + "edge_out_trans.mlp.0.linear.bias"]))
            deg = torch.zeros(N, device=dev).index_add_(0, rs["src"], torch.ones(E, device=dev)).clamp(min=1)
            aggt = torch.zeros(N, H, device=dev).index_add_(0, rs["src"], m) / deg[:, None]
        note(f"fwd layer {l} gcl edge: z1, ew', agg", rel(z1, tape.get(_capi.TAPE_Z1, l)[:E, :H]),
             rel(ew_new[:A], tape.get(_capi.TAPE_EW, l + 1)[:A, :W]), rel(aggt, tape.get(_capi.TAPE_AGG, l)[:, :H]))
        Gn = torch.randn(E, W, generator=gen).to(dev)
        if last:
            Gn[A:] = 0
        dagg = torch.randn(N, H, generator=gen).to(dev)
        gr = torch.autograd.grad([ew_new, aggt], [ew_t, Pt, Qt] + [P[n] for n in names], [Gn, dagg])
        dew = torch.zeros(E + 1, WP, device=dev)
        dew[:E, :W] = Gn
        dagg_p = torch.zeros(N, HP, device=dev)
        dagg_p[:, :H] = dagg
        dz3 = torch.zeros(E + 1, WP, device=dev)
        mout, dz2, dz1 = (torch.zeros(E + 1, HP, device=dev) for _ in range(3))
        da = torch.zeros(E + 1, device=dev)
        dPQ = torch.zeros(2, N, HP, device=dev)
        _capi.check(L.oard_gcl_backward_dx(C.byref(cfg), topo.handle, pbwd.data_ptr(), l, tape.buf.data_ptr(), dagg_p.data_ptr(),
                                           dew.data_ptr(), dz3.data_ptr(), mout.data_ptr(), dz2.data_ptr(), da.data_ptr(),
                                           dz1.data_ptr(), stream), "gcl bwd")
        _capi.check(L.oard_edge_node_sums(C.byref(cfg), topo.handle, dz1.data_ptr(), dPQ[0].data_ptr(), dPQ[1].data_ptr(), stream), "sums")
        note(f"bwd layer {l} gcl: dew, dP, dQ, m", rel(dew[:E, :W], gr[0]), rel(dPQ[0, :, :H], gr[1]), rel(dPQ[1, :, :H], gr[2]),
             rel(mout[:E, :H], m))
        rows3 = A if last else E
        gw, gb = training._wgrad(dz3, WP, W, W, W, mout, HP, False, H, H, H, rows3, True, dyn, stream)
        note(f"bwd layer {l} gcl: dW3, db3", rel(gw, gr[6]), rel(gb, gr[7]))
        gw, gb = training._wgrad(dz2, HP, H, H, H, tape.get(_capi.TAPE_Z1, l), HP, True, H, H, H, E, True, dyn, stream)
        note(f"bwd layer {l} gcl: dW2, db2", rel(gw, gr[4]), rel(gb, gr[5]))
        ewx = torch.zeros(E + 1, WP, device=dev)
        ewx[:E, :W] = ew_l
        gw, _ = training._wgrad(dz1, HP, H, H, H, ewx, WP, False, W, W, W, E, False, dyn, stream)
        note(f"bwd layer {l} gcl: dW1c", rel(gw, gr[3][:, 2 * H:]))
        m0t = F.silu(tape.get(_capi.TAPE_Z2, l)[:E, :H])
        note(f"bwd layer {l} gcl: dwatt, dbatt", rel((da[:E, None] * m0t).sum(0, keepdim=True), gr[8]), rel(da[:E].sum().reshape(1), gr[9]))
        # ---- EquiMessage gather half (k_equi_node_v1 part 1): HIP adjoint vs torch autograd ----
        if A > 0:
            with torch.no_grad():
                agg0 = tape.get(_capi.TAPE_AGG, l)[:, :H]
                s_mid, xq0 = training.stage_gcl_node(P, l, xh, agg0, H)
            xq_t = xq0.detach().clone().requires_grad_(True)
            cd_t = tape.get(_capi.TAPE_CD, l)[:A].view(A, 3, HP)[:, :, :H].detach().clone().requires_grad_(True)
            vec_t = tape.get(_capi.TAPE_VEC_IN, l).view(N, 3, HP)[:, :, :H].detach().clone().requires_grad_(True)
            rbfw = P[e + "rbf_proj.weight"]
            with torch.enable_grad():
                s_a, vec_a = training.stage_equi_message(P, l, s_mid, xq_t, cd_t, vec_t, g, H)
            note(f"fwd layer {l} equi message: s_a, vec_a", rel(s_a, tape.get(_capi.TAPE_S_A, l)[:, :H]),
                 rel(vec_a, tape.get(_capi.TAPE_VEC_A, l).view(N, 3, HP)[:, :, :H]))
            gS, gVc = torch.randn(N, H, generator=gen).to(dev), torch.randn(N, 3, H, generator=gen).to(dev)
            gm = torch.autograd.grad([s_a, vec_a], [xq_t, cd_t, vec_t, rbfw], [gS, gVc])
            gxh = (gS * training.INV_SQRT2).contiguous()
            crh = F.linear(g.rbf, rbfw).detach()
            dcdh, dcrh = torch.zeros(A + 1, 3, HP, device=dev), torch.zeros(A + 1, 3, HP, device=dev)
            dxqh, dvech = torch.empty(N, 3 * H, device=dev), torch.empty(N, 3, H, device=dev)
            _capi.check(L.oard_equi_msg_backward(C.byref(cfg), topo.handle, tape.buf.data_ptr(), l, xq0.contiguous().data_ptr(), crh.data_ptr(),
                                                 gxh.data_ptr(), gVc.contiguous().data_ptr(), dcdh.data_ptr(), dcrh.data_ptr(), dxqh.data_ptr(),
                                                 dvech.data_ptr(), stream), "equi msg bwd")
            grbf = training._wgrad(dcrh.view(A + 1, 3 * HP), 3 * HP, H, HP, 3 * H, tape.get(_capi.TAPE_RBF), training._pad16(R), False, R, R, R,
                                   A, False, dyn, stream)[0]
            note(f"bwd layer {l} equi message: dxq, dcd, dvec, drbf_proj", rel(dxqh, gm[0]), rel(dcdh[:A, :, :H], gm[1]), rel(dvech, gm[2]),
                 rel(grbf, gm[3]))
        # ---- EquiUpdate with the HIP frame-scalar op (Lin3uFunction) vs the plain torch formulation ----
        un = [n_ for n_ in P if n_.startswith(f"model.update_layers.{l}.")]
        sa0 = tape.get(_capi.TAPE_S_A, l)[:, :H]
        va0 = tape.get(_capi.TAPE_VEC_A, l).view(N, 3, HP)[:, :, :H]
        cs, cv = torch.randn(N, H, generator=gen).to(dev), torch.randn(N, 3, H, generator=gen).to(dev)
        res = []
        for hip in (None, (dyn, cfg, l, stream)):
            a_, b_ = sa0.detach().clone().requires_grad_(True), va0.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                o1, o2 = training.stage_equi_update(P, l, a_, b_, g, H, hip=hip)
            res.append((o1.detach(), o2.detach(), torch.autograd.grad([o1, o2], [a_, b_] + [P[n_] for n_ in un], [cs, cv])))
        note(f"fwd layer {l} equi update (HIP lin3u vs torch): s, vec", rel(res[1][0], res[0][0]), rel(res[1][1], res[0][1]))
        note(f"bwd layer {l} equi update (HIP lin3u vs torch)", *[rel(x_, y_) for x_, y_ in zip(res[1][2], res[0][2])])
        # ---- Equi edge ----
        if A > 0:
            ew1 = tape.get(_capi.TAPE_EW, l + 1)[:A, :W].detach().clone().requires_grad_(True)
            ns = [e + "dir_proj.0.weight", e + "dir_proj.0.bias", e + "dir_proj.2.weight", e + "dir_proj.2.bias"]
            with torch.enable_grad():
                cdt = F.linear(F.silu(F.linear(ew1, P[ns[0]], P[ns[1]])), P[ns[2]], P[ns[3]])
            note(f"fwd layer {l} equi edge: cd", rel(cdt.view(A, 3, H), tape.get(_capi.TAPE_CD, l)[:A].view(A, 3, HP)[:, :, :H]))
            dcd = torch.randn(A, 3 * H, generator=gen).to(dev)
            ge = torch.autograd.grad([cdt], [ew1] + [P[n] for n in ns], [dcd])
            dcd_p = torch.zeros(A + 1, 3, HP, device=dev)
            dcd_p[:A, :, :H] = dcd.view(A, 3, H)
            dew2 = torch.zeros(E + 1, WP, device=dev)
            dzd1 = torch.zeros(A + 1, D1P, device=dev)
            _capi.check(L.oard_equi_backward_dx(C.byref(cfg), topo.handle, pbwd.data_ptr(), l, tape.buf.data_ptr(), dcd_p.data_ptr(),
                                                dew2.data_ptr(), dzd1.data_ptr(), stream), "equi bwd")
            gw2, gb2 = training._wgrad(dcd_p.view(A + 1, 3 * HP), 3 * HP, H, HP, 3 * H, tape.get(_capi.TAPE_ZD1, l), D1P, True,
                                       3 * H, 3 * H, 3 * H, A, True, dyn, stream)
            gw0, gb0 = training._wgrad(dzd1, D1P, 3 * H, 3 * H, 3 * H, tape.get(_capi.TAPE_EW, l + 1), WP, False, W, W, W, A, True,
                                       dyn, stream)
            note(f"bwd layer {l} equi: dew, dp0 w b, dp2 w b", rel(dew2[:A, :W], ge[0]), rel(gw0, ge[1]), rel(gb0, ge[2]),
                 rel(gw2, ge[3]), rel(gb2, ge[4]))
    # ---- 3. whole step ---------------------------------------------------------------------------------------------------
    training.DynamicsFunction.forward = orig
    loss.backward()
    grads = {n: p.grad for n, p in dyn.named_parameters() if p.grad is not None}
    errs, flat = c.compare(grads)
    return out, errs, flat, c.meta["ref_f32_vs_f64"]


def tape_rows(topo, L, dev, stream):
    out = {}
    for key, which in (("src", _capi.TOPO_ROW_SRC), ("tgt", _capi.TOPO_ROW_TGT)):
        t = torch.empty(max(topo.E, 1), dtype=torch.int32, device=dev)
        _capi.check(L.oard_topology_export(topo.handle, which, t.data_ptr(), t.numel(), stream), "export")
        out[key] = t[: topo.E].long()
    return out


