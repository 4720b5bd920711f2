"""Stage-by-stage check of the training path on the GPU box (prints, asserts nothing):
  1. the torch stage functions of oareactdiff_amd/training.py recomputed from the taped inputs vs the taped outputs of the HIP forward;
  2. the HIP backward kernels of the two edge stages vs torch autograd on a torch restatement of the same stage, teacher-forced
     with the tape's inputs and random cotangents;
  3. the whole training step vs the reference gradient golden.
usage: python tools/grad_debug.py [case]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from _grad_cases import CNF, NODE_NFS, GradCase  # noqa: E402
from oareactdiff_amd import _capi, training  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))


def main(name):
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
    print(f"{name}: N {N} E {E} A {A} loss {float(loss):.8f} ref64 {float(c.z['f64_loss']):.8f}")
    geo = tape.get(_capi.TAPE_GEO)[:A]
    g = training.Geometry(topo.inner_src, topo.inner_tgt, topo.node_sample, topo.node_group, topo.B, topo.B * 3, geo,
                          tape.get(_capi.TAPE_RBF)[:A, :R], tape.get(_capi.TAPE_PP0)[:, 0], tape.get(_capi.TAPE_X1))
    stream = torch.cuda.current_stream(dev).cuda_stream
    L = _capi.lib()
    with torch.no_grad():
        # ---- 1. forward consistency of the stage functions -------------------------------------------------------
        hin = tape.get(_capi.TAPE_HIN)[:, :Cc]
        s0, ew0, c0 = training.stage_init(P, hin, g, H)
        print("init  s0", rel(s0, tape.get(_capi.TAPE_S_IN, 0)[:, :H]), " ew0", rel(ew0, tape.get(_capi.TAPE_EW, 0)[:A, :W]))
        for l in range(NL):
            s_in = tape.get(_capi.TAPE_S_IN, l)[:, :H]
            vec_in = tape.get(_capi.TAPE_VEC_IN, l).view(N, 3, HP)[:, :, :H]
            xh, Pn, Qn = training.stage_node_pre(P, l, s_in, g, H)
            agg = tape.get(_capi.TAPE_AGG, l)[:, :H]
            cd = tape.get(_capi.TAPE_CD, l)[:A].view(A, 3, HP)[:, :, :H]
            s_out, vec_out = training.stage_node_mid(P, l, xh, agg, cd, vec_in, g, H)
            print(f"layer {l}: s_out", rel(s_out, tape.get(_capi.TAPE_S_IN, l + 1)[:, :H]), " vec_out",
                  rel(vec_out, tape.get(_capi.TAPE_VEC_IN, l + 1).view(N, 3, HP)[:, :, :H]))
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
        print(f"layer {l}: torch edge stage vs tape: z1", rel(z1, tape.get(_capi.TAPE_Z1, l)[:E, :H]),
              " ew_new(inner)", rel(ew_new[:A], tape.get(_capi.TAPE_EW, l + 1)[:A, :W]),
              " agg", rel(aggt, tape.get(_capi.TAPE_AGG, l)[:, :H]))
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
        print(f"   GCL bwd: dew", rel(dew[:E, :W], gr[0]), " dP", rel(dPQ[0, :, :H], gr[1]), " dQ", rel(dPQ[1, :, :H], gr[2]),
              " m", rel(mout[:E, :H], m))
        rows3 = A if last else E
        gw, gb = training._wgrad(dz3, WP, W, W, W, mout, HP, False, H, H, H, rows3, True, dyn, stream)
        print("   dW3", rel(gw, gr[6]), " db3", rel(gb, gr[7]))
        gw, gb = training._wgrad(dz2, HP, H, H, H, tape.get(_capi.TAPE_Z1, l), HP, True, H, H, H, E, True, dyn, stream)
        print("   dW2", rel(gw, gr[4]), " db2", rel(gb, gr[5]))
        ewx = torch.zeros(E + 1, WP, device=dev)
        ewx[:E, :W] = ew_l
        gw, _ = training._wgrad(dz1, HP, H, H, H, ewx, WP, False, W, W, W, E, False, dyn, stream)
        print("   dW1c", rel(gw, gr[3][:, 2 * H:]))
        m0t = F.silu(tape.get(_capi.TAPE_Z2, l)[:E, :H])
        print("   dwatt", rel((da[:E, None] * m0t).sum(0, keepdim=True), gr[8]), " dbatt", rel(da[:E].sum().reshape(1), gr[9]))
        # ---- Equi edge ----
        if A > 0:
            ew1 = tape.get(_capi.TAPE_EW, l + 1)[:A, :W].detach().clone().requires_grad_(True)
            ns = [e + "dir_proj.0.weight", e + "dir_proj.0.bias", e + "dir_proj.2.weight", e + "dir_proj.2.bias"]
            with torch.enable_grad():
                cdt = F.linear(F.silu(F.linear(ew1, P[ns[0]], P[ns[1]])), P[ns[2]], P[ns[3]])
            print(f"   torch equi stage vs tape: cd", rel(cdt.view(A, 3, H), tape.get(_capi.TAPE_CD, l)[:A].view(A, 3, HP)[:, :, :H]))
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
            print("   Equi bwd: dew", rel(dew2[:A, :W], ge[0]), " ddp0", rel(gw0, ge[1]), rel(gb0, ge[2]), " ddp2", rel(gw2, ge[3]),
                  rel(gb2, ge[4]))
    # ---- 3. whole step ---------------------------------------------------------------------------------------------------
    loss.backward()
    grads = {n: p.grad for n, p in dyn.named_parameters() if p.grad is not None}
    errs, flat = c.compare(grads)
    gap = c.meta["ref_f32_vs_f64"]
    print(f"whole step: flat gradient error {flat:.2e}")
    for n in sorted(errs, key=lambda k: -errs[k]):
        print(f"  {n:62s} ours {errs[n]:.2e}   reference f32 {gap[n]:.2e}")


def tape_rows(topo, L, dev, stream):
    out = {}
    for key, which in (("src", _capi.TOPO_ROW_SRC), ("tgt", _capi.TOPO_ROW_TGT)):
        t = torch.empty(max(topo.E, 1), dtype=torch.int32, device=dev)
        _capi.check(L.oard_topology_export(topo.handle, which, t.data_ptr(), t.numel(), stream), "export")
        out[key] = t[: topo.E].long()
    return out


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "g9_grad_h32")
