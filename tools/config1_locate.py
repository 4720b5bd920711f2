"""Where does the error of config 1's worst teacher-forced network call live?  (round 6; the question of VERDICT r5 item 1)
Replays tests/test_configs.py::test_config1_single_reaction_t50_sampler's float64 trajectory (Gaussian-prior term, identical float32
inputs), ranks the 51 calls by the HIP path's velocity error, and for the worst ones prints every tap of the HIP path beside the SAME stage
of plain torch float32 (the oracle in float32 with float64 geometry and the exact node frame - the arithmetic the HIP path implements),
both against the float64 oracle.  A stage where the HIP column jumps and the torch column does not is a summation-order problem of ours.
usage (GPU box): python tools/config1_locate.py [n_worst] [OARD debug options as k=v ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import leftnet_oracle as oracle  # noqa: E402
import sampler_oracle as so  # noqa: E402
from _cases import rel  # noqa: E402
from oareactdiff_amd import _capi  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch  # noqa: E402
from oareactdiff_amd.sampler import DiffusionSampler  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402
from oareactdiff_amd.synthetic import make_inputs  # noqa: E402


def main():
    n_worst = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    L = _capi.lib()
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        assert L.oard_debug_option(k.encode(), int(v)) == 0, k
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=42)
    for k in list(sd):
        if "out_pos" in k and "update_net.2" in k:
            sd[k] = sd[k] * 0.05
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    dyn.nan_check = "async"
    B, nf, T = 1, 20, 50
    frag = [torch.tensor([nf]) for _ in range(3)]
    masks = [get_mask_for_frag(f) for f in frag]
    cm = torch.cat(masks)
    ei, nfs = get_edges_index(cm, remove_self_edge=True), get_n_frag_switch(frag)
    cond = torch.zeros(B, 1)
    h0 = [x[:, 3:].clone() for x in make_inputs(B, nf, masks, 5, "cpu")]
    gens = {}

    def noise(i):
        if i not in gens:
            g = torch.Generator().manual_seed(1000 + i)
            gens[i] = [torch.randn(nf, 9, generator=g) for _ in range(3)]
        return gens[i]
    smp = DiffusionSampler(dyn, "polynomial_2", T, 1e-5, pos_only=True, gaussian_prior_std=1.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    calls = []

    def dyn64(zt, t):
        o = oracle.dynamics_forward(sd64, cfg, zt, ei, t, cond.double(), nfs, cm, 1, nodeframe="exact")
        calls.append(([z.clone() for z in zt], t.clone()))
        c = smp.prior_coefficient(int(round(float(t.reshape(-1)[0]) * T)), T)
        return [torch.cat([x[:, :3] + c * z[:, :3], x[:, 3:]], dim=1) for x, z in zip(o, zt)]
    torch.set_default_dtype(torch.float64)
    table = so.gamma_table("polynomial_2", T, 1e-5).double()
    so.sample(dyn64, table, T, masks, B, lambda i: [n.double() for n in noise(i)], cond.double(), True, [h.double() for h in h0])
    torch.set_default_dtype(torch.float32)
    H, nl = cfg["hidden_channels"], cfg["num_layers"]
    inner = nfs[ei[0]] == nfs[ei[1]]
    rows = []
    for n, (zt, t) in enumerate(calls):
        z32 = [z.float() for z in zt]
        args = ([z.to(dev) for z in z32], ei.to(dev), t.float().to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
        with torch.no_grad():
            out, _ = dyn(*args)
        st64, st32 = {}, {}
        o64 = oracle.dynamics_forward(sd64, cfg, [z.double() for z in z32], ei, t.float().double(), cond.double(), nfs, cm, 1,
                                      nodeframe="exact", stages=st64)
        o32 = oracle.dynamics_forward(sd, cfg, z32, ei, t.float(), cond, nfs, cm, 1, nodeframe="exact", geom64=True, stages=st32)
        cat = lambda o: torch.cat([x[:, :3].cpu().double().reshape(-1) for x in o])       # noqa: E731
        rows.append((rel(cat(out), cat(o64)), rel(cat(o32), cat(o64)), n, args, st64, st32))
    print("call  hip_vel_err  torch32_vel_err")
    for e, e32, n, *_ in rows:
        print(f"{n:4d}  {e:.2e}    {e32:.2e}")
    print(f"max: hip {max(r[0] for r in rows):.2e}  torch32 {max(r[1] for r in rows):.2e};  median: hip "
          f"{sorted(r[0] for r in rows)[len(rows) // 2]:.2e}  torch32 {sorted(r[1] for r in rows)[len(rows) // 2]:.2e}")
    for e, e32, n, args, st64, st32 in sorted(rows, key=lambda r: -r[0])[:n_worst]:
        print(f"=== call {n}: hip {e:.2e} torch32 {e32:.2e}      stage: hip-vs-f64 | torch32-vs-f64 | max|ref|")

        def run(stop):
            L.oard_debug_stop_after(stop)
            with torch.no_grad():
                dyn(*args)
            torch.cuda.synchronize()
            L.oard_debug_stop_after(0)

        def show(tag, got, key, sel=None):
            want, t32 = st64[key], st32[key]
            got = got.detach().cpu().double().reshape(want.shape)
            if sel is not None:
                got, want, t32 = got[sel], want[sel], t32[sel]
            print(f"   {tag:24s} {rel(got, want):.2e} | {rel(t32, want):.2e} | {float(want.abs().max()):.2e}")
        run(1)
        show("s0", dyn.debug_tap(_capi.TAP_S), "s0")
        show("NE1", dyn.debug_tap(_capi.TAP_NE1), "NE1")
        show("edgeweight0", dyn.debug_tap(_capi.TAP_EDGE), "edgeweight0")
        for l in range(nl):
            run(100 + 10 * l + 1)
            show(f"l{l}.s_gcl", dyn.debug_tap(_capi.TAP_S), f"l{l}.s_gcl")
            show(f"l{l}.edgeweight[inner]", dyn.debug_tap(_capi.TAP_EDGE), f"l{l}.edgeweight", inner)
            run(100 + 10 * l + 2)
            show(f"l{l}.s", dyn.debug_tap(_capi.TAP_S), f"l{l}.s")
            show(f"l{l}.vec", dyn.debug_tap(_capi.TAP_VEC), f"l{l}.vec")
        run(0)
        show("dpos", dyn.debug_tap(_capi.TAP_DPOS), "dpos")
        show("h_out", dyn.debug_tap(_capi.TAP_HOUT), "h_out")
        # the output block alone: torch float64 on the HIP path's own s / vec (what the output kernel adds)
        s_hip, vec_hip = dyn.debug_tap(_capi.TAP_S).cpu().double(), dyn.debug_tap(_capi.TAP_VEC).cpu().double().reshape(-1, 3, H)
        o = "model.out_pos.output_network.0"
        v1 = torch.norm(vec_hip @ sd64[o + ".vec1_proj.weight"].t(), dim=-2)
        v2 = vec_hip @ sd64[o + ".vec2_proj.weight"].t()
        x = torch.cat([s_hip, v1], -1) @ sd64[o + ".update_net.0.weight"].t() + sd64[o + ".update_net.0.bias"]
        x = (x * torch.sigmoid(x)) @ sd64[o + ".update_net.2.weight"].t() + sd64[o + ".update_net.2.bias"]
        dpos_from_hip_state = (x[:, 1:2].unsqueeze(1) * v2).squeeze(-1)
        print(f"   output block in float64 on the HIP path's (s, vec): dpos {rel(dpos_from_hip_state, st64['dpos']):.2e} "
              f"(the kernel's own dpos from the same state: {rel(dyn.debug_tap(_capi.TAP_DPOS).cpu().double(), dpos_from_hip_state):.2e})")
        v2r = st64["l5.vec"] @ sd64[o + ".vec2_proj.weight"].t()
        print(f"   cancellation in vec2_proj: max|v2| {float(v2r.abs().max()):.2e} against sum|w||vec| "
              f"{float((st64['l5.vec'].abs() @ sd64[o + '.vec2_proj.weight'].abs().t()).max()):.2e}")


if __name__ == "__main__":
    main()
