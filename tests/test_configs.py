"""The configurations BASELINE.json names, at their real sizes, pinned on the reference / the float64 oracle:
  configs[1]  the launch bench.py times (B=64 x 3 x 23 atoms, L=6): golden reactions embedded in the batch;
  configs[0]  one ~20-atom reaction, T=50 `polynomial_2` sampling, production dims (device sampler vs float64 replay,
              every network call teacher-forced against the oracle);
  configs[4]  128-atom objects at the production depth L=6 (one reaction against the float64 oracle).
Plus the topology cache: entries keep their key tensors alive, so a recycled address can never alias another layout."""
import gc

import pytest
import torch

import leftnet_oracle as oracle
import sampler_oracle as so
from _cases import LIB_AUTO, Case, debug_options, rel
from oareactdiff_amd.graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
from oareactdiff_amd.synthetic import make_inputs, make_topology

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _prod_dynamics(dev, cfg=None, seed=42):
    from oareactdiff_amd.dynamics import EGNNDynamics
    cfg = dict(cfg or PRODUCTION_LEFTNET_CONFIG)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=seed)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    return dyn, sd, cfg


@pytest.mark.parametrize("parts", [0, 1])
def test_benched_launch_reproduces_the_golden_reactions(parts):
    """bench.py's launch (B=64, 23 atoms, L=6, throughput kernels; default 4-sub-batch schedule and parts=1) with the two
    reactions of golden g2 (reference float64 outputs) in slots 0, 1 and again in slots 62, 63."""
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    assert c.cfg["num_layers"] == 6 and c.xh[0].shape == (46, 9)
    B, nf = 64, 23
    with debug_options(parts=parts):
        dyn, _, _ = _prod_dynamics(dev, c.cfg)
        cm, nfs, ei, masks = make_topology(B, nf)
        xh = make_inputs(B, nf, masks, 99, "cpu")
        g = torch.Generator().manual_seed(1)
        t, cond = torch.rand(B, 1, generator=g), torch.rand(B, 1, generator=g)
        for slot0 in (0, 62):
            for k in range(3):
                xh[k][slot0 * nf:(slot0 + 2) * nf] = c.xh[k]
            t[slot0:slot0 + 2], cond[slot0:slot0 + 2] = c.t, c.conditions
        with torch.no_grad():
            out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    rv, rh = c.split(c.ref64)
    for slot0 in (0, 62):
        got = [o[slot0 * nf:(slot0 + 2) * nf].cpu() for o in out]
        v, h = c.split(got)
        print(f"parts={parts} slots {slot0},{slot0 + 1}: vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
        assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


def test_benched_launch_persistent_and_tile_gcl_kernels_agree():
    """The B = 64 launch bench.py times, as ONE sub-batch (the setting of `roofline`, where the persistent GCL kernel runs: 256
    workgroups, shares of 18 / 19 half-tiles) against the tile kernel: the same numbers to the last bits (one difference in the
    summation order of S1, csrc/oard_edge_p.h), and the persistent kernel bit-identical to itself on 97 workgroups."""
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    B, nf = 64, 23
    outs = []
    for persist, grid in ((0, 0), (1, 0), (1, 97)):
        with debug_options(parts=1, gcl_persist=persist, gcl_grid=grid):
            dyn, _, _ = _prod_dynamics(dev, c.cfg)
            cm, nfs, ei, masks = make_topology(B, nf)
            xh = make_inputs(B, nf, masks, 99, "cpu")
            g = torch.Generator().manual_seed(1)
            t, cond = torch.rand(B, 1, generator=g), torch.rand(B, 1, generator=g)
            with torch.no_grad():
                out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
            torch.cuda.synchronize()
            outs.append([o.clone() for o in out])
    for a, b, c2 in zip(*outs):
        assert bool(torch.isfinite(a).all()) and torch.equal(b, c2)
        assert (a - b).abs().max() <= 2e-6 * float(a.abs().max())


def test_config1_single_reaction_t50_sampler():
    """BASELINE configs[0]: B=1, 20 atoms per object, T=50 polynomial_2, production dims.  The device sampler follows a
    float64 replay of the same noise over the 51 network calls (accumulated tolerance 5e-5), and every single network
    call, fed with the float64 trajectory's state, is within 1e-5 of the float64 oracle ON IDENTICAL INPUTS.

    Round 6, two changes to what this test measures (the arithmetic under test did not change):
    * the trajectory.  With untrained weights eps_hat ~ 0 and the sampler multiplies its state by 1 / alpha_t|s per step: by the 4th
      call every pair is beyond the 10 A cutoff (|pos| ~ 100 ... 700 A), 47 of the 51 calls predicted an exactly-zero velocity and
      the worst call of the old test (9.05e-6 ... 1.03e-5) was the LAST one before that: 2 edges left inside the cutoff, |vel| ~ 2e-12.
      `gaussian_prior_std=1` (sampler.py) adds the ideal denoiser of N(0, 1) data to both the device loop and the float64 replay:
      every call now sees a molecule-sized cloud with all 1 140 same-object edges active;
    * identical inputs.  The float64 oracle used to see the float64 state and the device its float32 rounding.  On that old worst
      call the ROUNDING OF THE INPUTS ALONE moves the float64 oracle by 3.9e-5 (ulp(80 A) against 10 A - d in the cutoff envelope),
      while float32 arithmetic on identical inputs is 1.3e-6 off (measured with the oracle on the CPU).  The float64 replay now
      evaluates the network on the float32 rounding of its state - where the device, whose state IS float32, evaluates it - so every
      recorded call compares the two on identical inputs."""
    from oareactdiff_amd import DiffusionSampler
    dev = torch.device("cuda:0")
    dyn, sd, cfg = _prod_dynamics(dev)
    for k in list(sd):                                   # tame the untrained output head (as the g4 fixtures do)
        if "out_pos" in k and "update_net.2" in k:
            sd[k] = sd[k] * 0.05
    dyn.load_state_dict(sd, strict=True)
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
    smp.sample(B, frag, conditions=cond, h0=h0, noise_fn=noise)
    sd64 = {k: v.double() for k, v in sd.items()}
    calls = []

    def dyn64(zt, t):
        # the replay evaluates the network where the device evaluates it: on the float32 rounding of its (float64) state
        z32 = [z.float() for z in zt]
        t32 = t.float()
        st = {}
        o = oracle.dynamics_forward(sd64, cfg, [z.double() for z in z32], ei, t32.double(), cond.double(), nfs, cm, 1, nodeframe="exact",
                                    stages=st)
        calls.append((z32, t32, o, int(st["edge_mask"].sum()), [z.clone() for z in zt]))
        c = smp.prior_coefficient(int(round(float(t.reshape(-1)[0]) * T)), T)          # the device loop's coefficient
        return [torch.cat([x[:, :3] + c * z[:, :3], x[:, 3:]], dim=1) for x, z in zip(o, zt)]
    torch.set_default_dtype(torch.float64)
    try:
        table = so.gamma_table("polynomial_2", T, 1e-5).double()
        x64 = so.sample(dyn64, table, T, masks, B, lambda i: [n.double() for n in noise(i)], cond.double(), True,
                        [h.double() for h in h0])
    finally:
        torch.set_default_dtype(torch.float32)
    got = torch.cat([smp.last_x[k][:, :3].cpu().double().reshape(-1) for k in range(3)])
    want = torch.cat([x64[k][:, :3].reshape(-1) for k in range(3)])
    traj = rel(got, want)
    assert len(calls) == T + 1
    inner = int((nfs[ei[0]] == nfs[ei[1]]).sum())
    assert min(c[3] for c in calls) == inner, "the trajectory left the cutoff: the calls behind that point test nothing"
    per_call, rounding = [], []
    for z32, t32, o, _, zt in calls:                     # teacher-forced: the HIP network on the float64 trajectory's (rounded) state
        with torch.no_grad():
            out, _ = dyn([z.to(dev) for z in z32], ei.to(dev), t32.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
        v = torch.cat([x[:, :3].cpu().double().reshape(-1) for x in out])
        per_call.append(rel(v, torch.cat([x[:, :3].reshape(-1) for x in o])))
        rounding.append(max(rel(a.double(), b) for a, b in zip(z32, zt)))
    print(f"config 1: trajectory error {traj:.2e} over {T + 1} calls, {inner} of {inner} inner edges active in every call; per network "
          f"call on identical inputs max {max(per_call):.2e} median {sorted(per_call)[len(per_call) // 2]:.2e} "
          f"(float32 rounding of the state: {max(rounding):.1e} of its largest entry)")
    # measured in round 6 (MI355X): worst call 2.53e-6 under the suite's launch shapes, 2.55e-6 under the library's defaults
    # (OARD_TEST_SHAPES=auto); medians 1.41e-6 / 1.50e-6; plain torch float32 on the same inputs: worst 3.2e-6, median 1.4e-6.  On the way
    # there: round-5 kernels 6.0e-6 / 5.4e-6 (medians 3.0e-6 / 3.4e-6); with the two accumulations of csrc/oard_node_v1.h summed on their
    # own (k_node_pre_v1, k_neighbor_v1) 3.5e-6 / 4.7e-6; with the output head's update_net / vec2_proj in float64 (k_out_v1) the above.
    assert max(per_call) <= 5e-6                             # the hard bar is TOL = 1e-5: gated with a factor of two of margin
    assert sorted(per_call)[len(per_call) // 2] <= 3e-6
    assert traj <= 5e-5


def test_config5_one_reaction_at_production_depth():
    """BASELINE configs[4] at L=6: one 3 x 128-atom reaction (147,072 edges) against the float64 oracle."""
    dev = torch.device("cuda:0")
    dyn, sd, cfg = _prod_dynamics(dev, seed=7)
    nf = 128
    cm, nfs, ei, masks = make_topology(1, nf)
    xh = make_inputs(1, nf, masks, 23, "cpu")
    t, cond = torch.tensor([[0.37]]), torch.tensor([[0.0]])
    with torch.no_grad():
        out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    ref = oracle.dynamics_forward({k: v.double() for k, v in sd.items()}, cfg, [x.double() for x in xh], ei, t.double(),
                                  cond.double(), nfs, cm, 1, nodeframe="exact")
    v = torch.cat([o[:, :3].cpu().double().reshape(-1) for o in out])
    h = torch.cat([o[:, 3:].cpu().double().reshape(-1) for o in out])
    rv = torch.cat([o[:, :3].reshape(-1) for o in ref])
    rh = torch.cat([o[:, 3:].reshape(-1) for o in ref])
    print(f"config 5, L=6: vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
    assert rel(v, rv) <= TOL and rel(h, rh) <= TOL


def test_topology_cache_entries_pin_their_key_tensors():
    """Layouts with equal N and E but different sample sizes, given as int32 masks that are freed between the calls:
    each cached topology keeps (edge_index, n_frag_switch, combined_mask) alive, so the allocator cannot hand the same
    address to another layout, and every call agrees with a cache-free evaluation."""
    dev = torch.device("cuda:0")
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=1)
    dyn, sd, _ = _prod_dynamics(dev, cfg)

    def layout(sizes):
        natm = [torch.tensor(sizes) for _ in range(3)]
        cm = torch.cat([get_mask_for_frag(n) for n in natm])
        return cm, get_n_frag_switch(natm), get_edges_index(cm, remove_self_edge=True), natm

    def run(d, sizes, seed):
        cm, nfs, ei, natm = layout(sizes)
        masks = [get_mask_for_frag(n) for n in natm]
        g = torch.Generator().manual_seed(seed)
        xh = [torch.cat([torch.randn(m.numel(), 3, generator=g), torch.rand(m.numel(), 6, generator=g)], 1) for m in masks]
        cm32 = cm.to(dev, torch.int32)                  # int32: .to(int64) inside makes a copy, nothing of ours is viewed
        with torch.no_grad():
            out, _ = d([x.to(dev) for x in xh], ei.to(dev), torch.full((len(sizes), 1), 0.5, device=dev),
                       torch.zeros(len(sizes), 1, device=dev), nfs.to(dev), cm32)
        return [o.cpu() for o in out]
    results = []
    for sizes in ([3, 5, 4], [5, 4, 3], [4, 3, 5]):       # same N = 36, same E
        results.append(run(dyn, sizes, 3))
        gc.collect()
        torch.cuda.empty_cache()
    assert len(dyn._topo_cache) == 3
    for topo in dyn._topo_cache.values():
        assert len(topo.key_tensors) == 3
    for sizes, got in zip(([3, 5, 4], [5, 4, 3], [4, 3, 5]), results):
        fresh, _, _ = _prod_dynamics(dev, cfg)
        want = run(fresh, sizes, 3)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


@pytest.mark.parametrize("npb", [1, 2, 3, 4, 16])
def test_small_batch_node_shapes_reproduce_the_golden_reactions(npb):
    """The per-layer node kernels gather with the wave's 16 columns walking the ROWS when a workgroup holds <= 4 nodes
    (row_lanes: 16 / 8 / 4 lanes per node for npb = 1 / 2 / 3-4) and with one column per node otherwise.  B = 8 reactions
    under the library's launch heuristics (latency edge kernels) and every such npb, with the two reactions of golden g2
    (reference float64 outputs) in slots 0, 1 and 6, 7."""
    dev = torch.device("cuda:0")
    c = Case("g2_prod_b2_n23")
    B, nf = 8, 23
    with debug_options(**dict(LIB_AUTO, npb=npb)):
        dyn, _, _ = _prod_dynamics(dev, c.cfg)
        cm, nfs, ei, masks = make_topology(B, nf)
        xh = make_inputs(B, nf, masks, 5, "cpu")
        g = torch.Generator().manual_seed(2)
        t, cond = torch.rand(B, 1, generator=g), torch.rand(B, 1, generator=g)
        for slot0 in (0, 6):
            for k in range(3):
                xh[k][slot0 * nf:(slot0 + 2) * nf] = c.xh[k]
            t[slot0:slot0 + 2], cond[slot0:slot0 + 2] = c.t, c.conditions
        with torch.no_grad():
            out, _ = dyn([x.to(dev) for x in xh], ei.to(dev), t.to(dev), cond.to(dev), nfs.to(dev), cm.to(dev))
    rv, rh = c.split(c.ref64)
    for slot0 in (0, 6):
        got = [o[slot0 * nf:(slot0 + 2) * nf].cpu() for o in out]
        v, h = c.split(got)
        print(f"npb={npb} slots {slot0},{slot0 + 1}: vel {rel(v, rv):.2e} h {rel(h, rh):.2e}")
        assert rel(v, rv) <= TOL and rel(h, rh) <= TOL
