"""Sampling loop (row N1): the oracle restatement against the reference's own
EnVariationalDiffusion.sample (golden g4_*, replayed noise), the host schedule against the oracle's,
and — on the GPU — the on-device sampler against a float64 replay of the same trajectory."""
import json
import os

import numpy as np
import pytest
import torch

import leftnet_oracle as oracle
import sampler_oracle as so
from _cases import GOLDEN, rel
from oareactdiff_amd.graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch
from oareactdiff_amd.schedule import Schedule
from oareactdiff_amd.spec import state_spec, synthetic_state_dict

CASES = ["g4_sampler_posonly", "g4_sampler_full", "g4_sampler_full_prod"]     # _prod: H=196, R=96, L=6
ICASES = ["g5_inpaint", "g5_inpaint_prod"]


class SCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.meta = json.loads(str(z["meta"]))
        self.cfg, self.T, self.sizes, self.pos_only = (self.meta["model_config"], self.meta["T"], self.meta["sizes"],
                                                       self.meta["pos_only"])
        sd = synthetic_state_dict(state_spec(self.cfg, [9, 9, 9], 1), self.cfg, seed=42)
        for k in list(sd):
            if "out_pos" in k and "update_net.2" in k:
                sd[k] = sd[k] * self.meta["head_scale"]
        self.sd = sd
        self.frag = [torch.tensor(self.sizes) for _ in range(3)]
        self.masks = [get_mask_for_frag(f) for f in self.frag]
        self.cm = torch.cat(self.masks)
        self.ei = get_edges_index(self.cm, remove_self_edge=True)
        self.nfs = get_n_frag_switch(self.frag)
        self.B = len(self.sizes)
        self.cond = torch.zeros(self.B, 1)
        self.h0 = [torch.from_numpy(z[f"h0_{k}"]) for k in range(3)] if (self.pos_only and "h0_0" in z.files) else None
        self.noise = lambda i: [torch.from_numpy(z[f"noise{i}_{k}"]) for k in range(3)]
        self.table = torch.from_numpy(z["table"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_sampler_replays_reference_bitwise(name):
    c = SCase(name)

    def dyn(zt, t):
        return oracle.dynamics_forward(c.sd, c.cfg, zt, c.ei, t, c.cond, c.nfs, c.cm, 1, nodeframe="literal",
                                       direct_vel=False)

    x = so.sample(dyn, c.table, c.T, c.masks, c.B, c.noise, c.cond, c.pos_only, c.h0)
    for k in range(3):
        assert torch.equal(x[k][:, :3], torch.from_numpy(c.z[f"ref_pos{k}"]))       # reference sampler output
        assert torch.equal(x[k], torch.from_numpy(c.z[f"oracle_x{k}"]))


def test_host_schedule_matches_oracle_and_fixture():
    for name, T, prec in (("polynomial_2", 20, 1e-5), ("polynomial_2", 1000, 1e-5), ("cosine", 100, 1e-5)):
        s = Schedule(name, T, prec)
        assert torch.equal(s.gamma, so.gamma_table(name, T, prec))
    c = SCase("g4_sampler_posonly")
    assert torch.equal(Schedule("polynomial_2", c.T, c.meta["precision"]).gamma, c.table)
    s = Schedule("polynomial_2", 1000, 1e-5)
    co = s.step(499)
    assert 0.9 < co.alpha_ts <= 1.0 and co.sigma > 0 and co.c_eps > 0
    assert s.index(500, 1000) == 500 and s.index(3, 10) == 300


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_sampler_tracks_f64_replay(name):
    from oareactdiff_amd import DiffusionSampler, EGNNDynamics
    dev = torch.device("cuda:0")
    c = SCase(name)
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(c.sd, strict=True)
    smp = DiffusionSampler(dyn, "polynomial_2", c.T, c.meta["precision"], pos_only=c.pos_only)
    out, masks = smp.sample(c.B, c.frag, conditions=c.cond, h0=c.h0, noise_fn=c.noise)
    assert int(smp.last_status[0].item()) == 0
    # float64 replay of the same noise: float64 oracle dynamics (exact node frame), float64 sampler arithmetic
    sd64 = {k: v.double() for k, v in c.sd.items()}

    def dyn64(zt, t):
        return oracle.dynamics_forward(sd64, c.cfg, zt, c.ei, t, c.cond.double(), c.nfs, c.cm, 1, nodeframe="exact")

    torch.set_default_dtype(torch.float64)
    try:
        x64 = so.sample(dyn64, c.table.double(), c.T, c.masks, c.B, lambda i: [n.double() for n in c.noise(i)],
                        c.cond.double(), c.pos_only, [h.double() for h in c.h0] if c.h0 else None)
    finally:
        torch.set_default_dtype(torch.float32)
    got = torch.cat([smp.last_x[k][:, :3].cpu().double().reshape(-1) for k in range(3)])
    want = torch.cat([x64[k][:, :3].reshape(-1) for k in range(3)])
    e = rel(got, want)
    print(f"{name}: device sampler vs float64 replay, positions rel = {e:.2e} over {c.T + 1} network calls")
    assert e <= 5e-5
    if not c.pos_only:
        goth = torch.cat([smp.last_x[k][:, 3:].cpu().double().reshape(-1) for k in range(3)])
        wanth = torch.cat([x64[k][:, 3:].reshape(-1) for k in range(3)])
        assert rel(goth, wanth) <= 5e-5
    # returned structure as en_diffusion.py:554-560
    assert len(out) == 1 and len(out[0]) == 3 and out[0][0].shape == (sum(c.sizes), 9)
    assert all(torch.equal(m.cpu(), mm) for m, mm in zip(masks, c.masks))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_graphed_sampler_is_bitwise_identical_to_the_eager_loop(name):
    """Small batches replay ONE captured hipGraph per step (device-side step counter, schedule / noise tables): same kernels,
    same arithmetic, so the samples must be bit-identical to the eager loop's - with injected noise and with the device RNG."""
    from oareactdiff_amd import DiffusionSampler, EGNNDynamics
    dev = torch.device("cuda:0")
    c = SCase(name)
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(c.sd, strict=True)
    smp = DiffusionSampler(dyn, "polynomial_2", c.T, c.meta["precision"], pos_only=c.pos_only)
    res = {}
    for graph in (False, True):
        out, _ = smp.sample(c.B, c.frag, conditions=c.cond, h0=c.h0, noise_fn=c.noise, graph=graph)
        res[graph] = ([x.clone() for x in smp.last_x], [o.clone() for o in out[0]])
    for a, b in zip(res[False][0] + res[False][1], res[True][0] + res[True][1]):
        assert torch.equal(a, b)
    rng = {}
    for graph in (False, True):
        torch.manual_seed(77)
        smp.sample(c.B, c.frag, conditions=c.cond, h0=c.h0, graph=graph)
        rng[graph] = [x.clone() for x in smp.last_x]
    for a, b in zip(rng[False], rng[True]):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    # the pre-drawn noise is bounded: with room for 3 steps per block the table is refilled between replays, same result
    smp.noise_block_bytes = 3 * 4 * sum(int(x.numel()) for x in smp.last_x)
    torch.manual_seed(77)
    smp.sample(c.B, c.frag, conditions=c.cond, h0=c.h0, graph=True)
    for a, b in zip(rng[False], smp.last_x):
        assert torch.equal(a, b)
    out, _ = smp.sample(c.B, c.frag, conditions=c.cond, h0=c.h0, noise_fn=c.noise, graph=True)
    for a, b in zip(res[False][0], smp.last_x):
        assert torch.equal(a, b)
    # ... and with the Gaussian-prior term of the benchmark runs (round 6): both loops take the coefficient from the same device table
    prior = DiffusionSampler(dyn, "polynomial_2", c.T, c.meta["precision"], pos_only=c.pos_only, gaussian_prior_std=1.0)
    got = {}
    for graph in (False, True):
        prior.sample(c.B, c.frag, conditions=c.cond, h0=c.h0, noise_fn=c.noise, graph=graph)
        got[graph] = [x.clone() for x in prior.last_x]
    for a, b, plain in zip(got[False], got[True], res[False][0]):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
        if a.numel():
            assert not torch.equal(a, plain)                    # the term does something


class ICase(SCase):
    def __init__(self, name="g5_inpaint"):
        super().__init__(name)
        self.xh_fixed = [torch.from_numpy(self.z[f"xh_fixed{k}"]) for k in range(3)]
        self.frag_fixed = self.meta["frag_fixed"]
        self.res, self.jump = self.meta["resamplings"], self.meta["jump_length"]


@pytest.mark.parametrize("name", ICASES)
def test_oracle_inpaint_replays_reference_bitwise(name):
    from oareactdiff_amd.schedule import get_repaint_schedule
    c = ICase(name)

    def dyn(zt, t):
        return oracle.dynamics_forward(c.sd, c.cfg, zt, c.ei, t, c.cond, c.nfs, c.cm, 1, nodeframe="literal",
                                       direct_vel=False)

    x = so.inpaint(dyn, c.table, c.T, c.masks, c.B, c.noise, c.cond, True, c.xh_fixed, c.frag_fixed, c.res, c.jump)
    for k in range(3):
        assert torch.equal(x[k][:, :3], torch.from_numpy(c.z[f"ref_pos{k}"]))       # reference inpaint output
    assert get_repaint_schedule(c.res, c.jump, c.T) == so.get_repaint_schedule(c.res, c.jump, c.T)
    # sum(out) - (len(out) - 1) * jump_length == timesteps   (_schedule.py:210)
    for r, j, t in ((2, 3, 12), (5, 5, 150), (10, 10, 250), (1, 1, 7)):
        sch = get_repaint_schedule(r, j, t)
        n_jumps = len(sch) - 1
        assert sum(sch) - n_jumps * j == t


@pytest.mark.gpu
@pytest.mark.parametrize("name", ICASES)
def test_device_inpaint_tracks_f64_replay(name):
    from oareactdiff_amd import DiffusionSampler, EGNNDynamics
    dev = torch.device("cuda:0")
    c = ICase(name)
    dyn = EGNNDynamics(model_config=dict(c.cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(c.sd, strict=True)
    smp = DiffusionSampler(dyn, "polynomial_2", c.T, c.meta["precision"], pos_only=True)
    out, masks = smp.inpaint(c.B, c.frag, conditions=c.cond, resamplings=c.res, jump_length=c.jump,
                             xh_fixed=[x.clone() for x in c.xh_fixed], frag_fixed=c.frag_fixed, noise_fn=c.noise)
    assert int(smp.last_status[0].item()) == 0
    sd64 = {k: v.double() for k, v in c.sd.items()}

    def dyn64(zt, t):
        return oracle.dynamics_forward(sd64, c.cfg, zt, c.ei, t, c.cond.double(), c.nfs, c.cm, 1, nodeframe="exact")

    torch.set_default_dtype(torch.float64)
    try:
        x64 = so.inpaint(dyn64, c.table.double(), c.T, c.masks, c.B, lambda i: [n.double() for n in c.noise(i)],
                         c.cond.double(), True, [x.double() for x in c.xh_fixed], c.frag_fixed, c.res, c.jump)
    finally:
        torch.set_default_dtype(torch.float32)
    got = torch.cat([smp.last_x[k][:, :3].cpu().double().reshape(-1) for k in range(3)])
    want = torch.cat([x64[k][:, :3].reshape(-1) for k in range(3)])
    e = rel(got, want)
    print(f"{name}: device inpaint vs float64 replay, positions rel = {e:.2e} ({c.meta['ncalls']} noise draws)")
    assert e <= 5e-5
    assert out[0][1].shape == (sum(c.sizes), 9)
