"""Generate tests/golden/g4_sampler.npz from the reference's EnVariationalDiffusion.sample.
BUILD-CONTAINER ONLY (imports /root/reference).  Records every torch.randn draw of the reference run so
the trajectory can be replayed by the oracle and by the HIP sampler."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oa_reactdiff.diffusion._normalizer import Normalizer  # noqa: E402
from oa_reactdiff.diffusion._schedule import DiffSchedule, PredefinedNoiseSchedule  # noqa: E402
from oa_reactdiff.diffusion.en_diffusion import EnVariationalDiffusion  # noqa: E402
from oa_reactdiff.dynamics import EGNNDynamics  # noqa: E402
from oa_reactdiff.model import LEFTNet  # noqa: E402

import leftnet_oracle as oracle  # noqa: E402
import sampler_oracle as so  # noqa: E402
from oareactdiff_amd.spec import state_spec, synthetic_state_dict  # noqa: E402


def run(name, pos_only, T, sizes, cfg, schedule_name="polynomial_2", precision=1e-5):
    node_nfs, cnf = [9, 9, 9], 1
    sd = synthetic_state_dict(state_spec(cfg, node_nfs, cnf), cfg, seed=42)
    # scale the output head down so that the untrained net is a mild perturbation and the trajectory stays bounded
    for k in list(sd):
        if "out_pos" in k and "update_net.2" in k:
            sd[k] = sd[k] * 0.1
    torch.set_default_dtype(torch.float32)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=node_nfs, edge_nf=0,
                       condition_nf=cnf, model=LEFTNet, device=torch.device("cpu"))
    dyn.load_state_dict(sd, strict=True)
    dyn.eval()
    norm = Normalizer((1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 3)
    gm = PredefinedNoiseSchedule(schedule_name, T, precision)
    ddpm = EnVariationalDiffusion(dynamics=dyn, schdule=DiffSchedule(gm, (1.0, 1.0, 1.0)), normalizer=norm,
                                  size_histogram=None, loss_type="l2", pos_only=pos_only)
    B = len(sizes)
    frag = [torch.tensor(sizes) for _ in range(3)]
    cond = torch.zeros(B, 1)
    h0 = None
    if pos_only:
        g = torch.Generator().manual_seed(5)
        h0 = []
        for k in range(3):
            n = sum(sizes)
            typ = torch.randint(0, 4, (n,), generator=g)
            f = torch.zeros(n, 6)
            f[torch.arange(n), typ] = 1.0
            f[:, 5] = torch.tensor([1.0, 6.0, 7.0, 8.0])[typ]
            h0.append(f)
    rec = []
    real_randn = torch.randn

    def spy(*a, **kw):
        x = real_randn(*a, **kw)
        rec.append(x.clone())
        return x

    torch.manual_seed(0)
    torch.randn = spy
    try:
        out, masks = ddpm.sample(n_samples=B, fragments_nodes=frag, conditions=cond, return_frames=1, timesteps=None, h0=h0)
    finally:
        torch.randn = real_randn
    # draws come in (pos, feat) pairs per object, per noise call
    assert len(rec) == (T + 2) * 3 * 2
    calls = [[torch.cat([rec[(c * 3 + k) * 2], rec[(c * 3 + k) * 2 + 1]], dim=1) for k in range(3)] for c in range(T + 2)]

    # replay with the oracle: dynamics = oracle restatement (float32, reference arithmetic)
    table = so.gamma_table(schedule_name, T, precision)
    assert torch.equal(table, gm.gamma.data)
    cm = torch.cat(masks)
    from oareactdiff_amd.graph_tools import get_edges_index, get_n_frag_switch
    ei = get_edges_index(cm, remove_self_edge=True)
    nfs = get_n_frag_switch(frag)

    def odyn(z, t):
        return oracle.dynamics_forward(sd, cfg, z, ei, t, cond, nfs, cm, cnf, nodeframe="literal", direct_vel=False)

    trace = []
    x = so.sample(odyn, table, T, masks, B, lambda i: calls[i], cond, pos_only, h0, trace=trace)
    ref_pos = [o[:, :3] for o in out[0]]
    err = max(float((x[k][:, :3] - ref_pos[k]).abs().max()) for k in range(3))
    scale = max(float(r.abs().max()) for r in ref_pos)
    print(name, "oracle sampler vs reference sampler: max|dpos| =", err, "scale", scale)
    arrays = {"table": table.numpy(), "meta": np.array(json.dumps(dict(name=name, pos_only=pos_only, T=T, sizes=sizes,
                                                                         model_config=cfg, schedule=schedule_name,
                                                                         precision=precision, head_scale=0.1,
                                                                         oracle_vs_ref_abs=err, scale=scale)))}
    for c in range(T + 2):
        for k in range(3):
            arrays[f"noise{c}_{k}"] = calls[c][k].numpy()
    for k in range(3):
        arrays[f"ref_pos{k}"] = ref_pos[k].numpy()
        arrays[f"ref_cat{k}"] = out[0][k][:, 3:-1].numpy()
        arrays[f"ref_charge{k}"] = out[0][k][:, -1:].numpy()
        arrays[f"oracle_x{k}"] = x[k].numpy()
        arrays[f"z_mid{k}"] = trace[T // 2][k].numpy()
        if h0 is not None:
            arrays[f"h0_{k}"] = h0[k].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **arrays)


def run_inpaint(name, T, sizes, cfg, resamplings, jump_length, frag_fixed, schedule_name="polynomial_2", precision=1e-5):
    node_nfs, cnf = [9, 9, 9], 1
    sd = synthetic_state_dict(state_spec(cfg, node_nfs, cnf), cfg, seed=42)
    for k in list(sd):
        if "out_pos" in k and "update_net.2" in k:
            sd[k] = sd[k] * 0.1
    torch.set_default_dtype(torch.float32)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=node_nfs, edge_nf=0,
                       condition_nf=cnf, model=LEFTNet, device=torch.device("cpu"))
    dyn.load_state_dict(sd, strict=True)
    dyn.eval()
    gm = PredefinedNoiseSchedule(schedule_name, T, precision)
    ddpm = EnVariationalDiffusion(dynamics=dyn, schdule=DiffSchedule(gm, (1.0, 1.0, 1.0)),
                                  normalizer=Normalizer((1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 3), size_histogram=None,
                                  loss_type="l2", pos_only=True)
    B = len(sizes)
    frag = [torch.tensor(sizes) for _ in range(3)]
    cond = torch.zeros(B, 1)
    g = torch.Generator().manual_seed(9)
    n = sum(sizes)
    xh_fixed = []
    for k in range(3):
        typ = torch.randint(0, 4, (n,), generator=g)
        f = torch.zeros(n, 6)
        f[torch.arange(n), typ] = 1.0
        f[:, 5] = torch.tensor([1.0, 6.0, 7.0, 8.0])[typ]
        xh_fixed.append(torch.cat([torch.randn(n, 3, generator=g), f], dim=1))
    rec = []
    real_randn = torch.randn

    def spy(*a, **kw):
        x = real_randn(*a, **kw)
        rec.append(x.clone())
        return x

    torch.manual_seed(1)
    torch.randn = spy
    try:
        out, masks = ddpm.inpaint(n_samples=B, fragments_nodes=frag, conditions=cond, return_frames=1,
                                  resamplings=resamplings, jump_length=jump_length, timesteps=None,
                                  xh_fixed=[x.clone() for x in xh_fixed], frag_fixed=frag_fixed)
    finally:
        torch.randn = real_randn
    ncalls = len(rec) // 6
    assert len(rec) == ncalls * 6
    calls = [[torch.cat([rec[(c * 3 + k) * 2], rec[(c * 3 + k) * 2 + 1]], dim=1) for k in range(3)] for c in range(ncalls)]
    table = so.gamma_table(schedule_name, T, precision)
    cm = torch.cat(masks)
    from oareactdiff_amd.graph_tools import get_edges_index, get_n_frag_switch
    ei = get_edges_index(cm, remove_self_edge=True)
    nfs = get_n_frag_switch(frag)

    def odyn(z, t):
        return oracle.dynamics_forward(sd, cfg, z, ei, t, cond, nfs, cm, cnf, nodeframe="literal", direct_vel=False)

    x = so.inpaint(odyn, table, T, masks, B, lambda i: calls[i], cond, True, xh_fixed, frag_fixed, resamplings, jump_length)
    ref_pos = [o[:, :3] for o in out[0]]
    err = max(float((x[k][:, :3] - ref_pos[k]).abs().max()) for k in range(3))
    scale = max(float(r.abs().max()) for r in ref_pos)
    print(name, "oracle inpaint vs reference inpaint: max|dpos| =", err, "scale", scale, "noise calls", ncalls)
    arrays = {"table": table.numpy(), "meta": np.array(json.dumps(dict(
        name=name, pos_only=True, T=T, sizes=sizes, model_config=cfg, schedule=schedule_name, precision=precision,
        head_scale=0.1, resamplings=resamplings, jump_length=jump_length, frag_fixed=frag_fixed, ncalls=ncalls,
        oracle_vs_ref_abs=err, scale=scale)))}
    for c in range(ncalls):
        for k in range(3):
            arrays[f"noise{c}_{k}"] = calls[c][k].numpy()
    for k in range(3):
        arrays[f"xh_fixed{k}"] = xh_fixed[k].numpy()
        arrays[f"ref_pos{k}"] = ref_pos[k].numpy()
        arrays[f"oracle_x{k}"] = x[k].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **arrays)


if __name__ == "__main__":
    if "--prod" in sys.argv:
        # production dims (H=196, R=96, L=6; train_ts1x.py:43-56) on small reactions: the device loops at the kernel
        # instantiations the bench runs.  pos_only=False exercises the feature half of the sampler step as well.
        from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG
        pcfg = dict(PRODUCTION_LEFTNET_CONFIG)
        run("g4_sampler_full_prod", False, 8, [10, 12], pcfg)
        run_inpaint("g5_inpaint_prod", 8, [10, 11], pcfg, resamplings=2, jump_length=2, frag_fixed=[0, 2])
        sys.exit(0)
    cfg = dict(pos_require_grad=False, cutoff=10.0, num_layers=2, hidden_channels=32, num_radial=8, in_hidden_channels=8)
    run("g4_sampler_posonly", True, 20, [4, 6], cfg)
    run("g4_sampler_full", False, 12, [5, 3], cfg)
    run_inpaint("g5_inpaint", 12, [4, 5], cfg, resamplings=2, jump_length=3, frag_fixed=[0, 2])
