"""bench.py — denoising-step throughput of the HIP path on synthetic 23-atom R/TS/P triples.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one EGNNDynamics.forward call (= one denoising step of the T=1000 sampler) on a batch of
64 reactions per GPU (BASELINE.json configs[1]); reactions are independent, so ranks run replicas with
no data-path collective (weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this driver)

import torch  # noqa: E402

H, R, L = 196, 96, 6
W = 3 * H + R
# MACs per edge per layer executed by the two hot kernels, as the reference formulates them
#  minus the node-only part of edge_mlp.0 that k_node_pre evaluates per node
MAC_GCL_S1, MAC_GCL_S2, MAC_GCL_S3 = W * H, H * H + H, H * W   # stages of k_gcl_edge: W1c.ew | W2, gate | W3 (residual)
MAC_GCL_EDGE = MAC_GCL_S1 + MAC_GCL_S2 + MAC_GCL_S3  # 306,740  (k_gcl_edge; reference form incl. 2H*H: 383,572)
MAC_EQUI_EDGE = 3 * R * H + 3 * H * W + 9 * H * H    # 804,384  (k_equi_edge)
PEAK_F32_MFMA = 157.3e12


def cpu_baseline(n_atoms: int, threads: int):
    """Times the CPU oracle (dense reference formulation, float32) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import leftnet_oracle as oracle
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    from oareactdiff_amd.synthetic import make_inputs, make_topology
    threads = max(1, min(threads, 16))      # the eager formulation stops scaling (and thrashes) beyond ~16 threads
    torch.set_num_threads(threads)
    cfg = dict(PRODUCTION_LEFTNET_CONFIG)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)
    B = 4
    cm, nfs, ei, masks = make_topology(B, n_atoms)
    xh = make_inputs(B, n_atoms, masks, 7, "cpu")
    t = torch.full((B, 1), 0.5)
    cond = torch.zeros(B, 1)
    with torch.no_grad():                                   # warm-up
        oracle.dynamics_forward(sd, cfg, xh, ei, t, cond, nfs, cm, 1, nodeframe="literal")
    calls, t0 = 0, time.perf_counter()
    while calls < 3 or (time.perf_counter() - t0 < 12.0 and calls < 40):      # ~12-20 s of CPU work
        with torch.no_grad():
            oracle.dynamics_forward(sd, cfg, xh, ei, t, cond, nfs, cm, 1, nodeframe="literal")
        calls += 1
    dt = time.perf_counter() - t0
    return {"value": B * calls / dt, "unit": "reaction-steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle/leftnet_oracle.py, float32, B={B} x {n_atoms}-atom triples, {calls} calls after 1 warm-up, "
                      f"{dt:.1f} s"}


def pmc_traffic(kernel: str, batch: int, atoms: int):
    """HBM bytes per denoising step of all launches of `kernel`, from the committed PMC passes
    (profiles/round1_pmc_{fetch,write}.txt, collected at the default workload): FETCH_SIZE (KiB; x2 — it
    under-reports wide streaming reads by 2x on gfx950, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KiB).
    None for any other workload."""
    if (batch, atoms) != (64, 23):
        return None
    import re
    vals = {}
    for key, fn in (("FETCH_SIZE", "round1_pmc_fetch.txt"), ("WRITE_SIZE", "round1_pmc_write.txt")):
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            return None
        cur = None
        for line in open(path):
            if line.startswith("=="):
                cur = line
            m = re.search(key + r"\s+per forward = ([0-9.e+]+)", line)
            if m and cur and kernel in cur:
                vals[key] = float(m.group(1))
    if len(vals) != 2:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="reactions per GPU")
    ap.add_argument("--atoms", type=int, default=23, help="atoms per object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dist = None
    backend = os.environ.get("OARD_BENCH_BACKEND", "nccl")    # "gloo" only to exercise the N>1 code path on a 1-GPU box
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from oareactdiff_amd import _capi
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    from oareactdiff_amd.synthetic import make_inputs, make_topology

    cfg = dict(PRODUCTION_LEFTNET_CONFIG)
    B, nf = args.batch, args.atoms
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg), strict=True)
    dyn.nan_check = "async"                 # no host sync inside the step
    cm, nfs, ei, masks = make_topology(B, nf)
    cm, nfs, ei = cm.to(dev), nfs.to(dev), ei.to(dev)
    # a few fixed-distribution input sets, resident in HBM, cycled per step (fresh noise each step)
    inputs = [make_inputs(B, nf, masks, 1234 + 17 * rank + k, dev) for k in range(4)]
    cond = torch.zeros(B, 1, device=dev)
    T = 1000
    ts = [torch.full((B, 1), (T - s) / T, device=dev) for s in range(8)]

    def step(i):
        with torch.no_grad():
            dyn(inputs[i % len(inputs)], ei, ts[i % len(ts)], cond, nfs, cm)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if not os.environ.get("OARD_BENCH_ALLOW_NAN"):          # (kernel ablation experiments produce garbage on purpose)
        assert int(dyn.last_status[0].item()) == 0, "NaN in the timed region"

    # per-kernel durations (HIP events on the launch stream), outside the timed region
    E = B * 3 * nf * (3 * nf - 1)
    A = B * 3 * nf * (nf - 1)
    roof = None
    if rank == 0:
        # Kernels are timed in isolation on the WHOLE batch (one sub-batch, nothing overlapping): that is the
        # figure a roofline fraction is about.  The timed region above runs the default schedule (4 concurrent
        # sub-batches), which is faster than the sum of the isolated kernels.
        L_ = _capi.lib()
        L_.oard_debug_option(b"parts", 1)
        dyn._topo_cache.clear()
        step(0)
        torch.cuda.synchronize(dev)
        L_.oard_timing_reset()
        L_.oard_timing_enable(1)
        for i in range(3):
            step(i)
        torch.cuda.synchronize(dev)
        L_.oard_timing_enable(0)
        L_.oard_debug_option(b"parts", int(os.environ.get("OARD_PARTS", "0")))
        dyn._topo_cache.clear()
        fam = {}
        for f in ("gcl_edge", "equi_edge", "node", "init", "other"):
            ms, n = _capi.timing_get(f)
            fam[f] = {"avg_ms": ms / max(n, 1), "launches_per_step": n / 3, "ms_per_step": ms / 3}
        # algorithmic FLOPs per STEP of each hot kernel family (L launches of the Equi kernel; the GCL kernel runs
        # once per layer on all edges, except that on inter-object edges the first layer has no S1 (constant
        # initial state: exact structural shortcut) and the last layer no S3 (its result is never read)
        flops = {"gcl_edge": 2.0 * (L * MAC_GCL_EDGE * E - (MAC_GCL_S1 + MAC_GCL_S3) * (E - A)),
                 "equi_edge": 2.0 * L * MAC_EQUI_EDGE * A}
        dom = max(flops, key=lambda f: fam[f]["ms_per_step"])
        oth = [f for f in flops if f != dom][0]
        ach = flops[dom] / (fam[dom]["ms_per_step"] * 1e-3)
        roof = {"bound": "mfma", "kernel": "k_" + dom, "achieved": ach / 1e12, "peak": PEAK_F32_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA, "traffic": pmc_traffic("k_" + dom, B, nf),
                "traffic_note": "HBM bytes per step over all launches of this kernel (committed PMC passes)",
                "algorithmic_flops_per_step": flops[dom], "launches_per_step": fam[dom]["launches_per_step"],
                "kernel_ms_per_step": fam[dom]["ms_per_step"], "avg_launch_ms": fam[dom]["avg_ms"],
                "families_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in fam.items()},
                "other_kernel": {"kernel": "k_" + oth, "achieved": flops[oth] / (fam[oth]["ms_per_step"] * 1e-3) / 1e12,
                                 "algorithmic_flops_per_step": flops[oth], "kernel_ms_per_step": fam[oth]["ms_per_step"]}}

    # the real sampling loop (row N1): T_probe genuine ancestral steps (network + fused sampler kernel + RNG)
    sampler_leg = None
    if rank == 0:
        from oareactdiff_amd.sampler import DiffusionSampler
        T_probe = 12
        smp = DiffusionSampler(dyn, "polynomial_2", T_probe, 1e-5, pos_only=True)
        frag = [torch.full((B,), nf, dtype=torch.long) for _ in range(3)]
        h0 = [x[:, 3:].clone() for x in inputs[0]]
        smp.sample(B, frag, conditions=cond, h0=h0)                      # warm-up (topology, buffers)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        smp.sample(B, frag, conditions=cond, h0=h0)
        torch.cuda.synchronize(dev)
        dts = time.perf_counter() - t1
        per_call = dts / (T_probe + 1)
        sampler_leg = {"ms_per_network_call_incl_sampler_step": per_call * 1e3, "network_calls": T_probe + 1,
                       "reactions_per_sec_T1000_projected": B / (1001 * per_call)}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": "denoising_steps_per_sec", "value": value, "unit": "reaction-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"EGNNDynamics.forward (LEFTNet H=196 R=96 L=6), B={B} reactions/GPU x 3 objects x "
                                   f"{nf} atoms, complete graph per reaction (N={B * 3 * nf}, E={E}), T=1000 sampler step shape",
                       "batch_per_gpu": B, "atoms_per_object": nf, "parallelism": f"replica x{world} (no collective)"},
            "batch_steps_per_sec_per_gpu": args.steps / dt,
            "reactions_per_sec_T1000": value / 1001.0,
            "roofline": roof,
            "sampler_loop": sampler_leg,
        }
        if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(nf, os.cpu_count() or 1)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
