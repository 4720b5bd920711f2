"""bench.py — denoising-step throughput of the HIP path on synthetic 23-atom R/TS/P triples.

  python bench.py --gpus N --steps K --warmup W
  N > 1, either form (one process per GPU, backend nccl = RCCL):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...      (no WORLD_SIZE in the environment: bench.py starts that launcher itself as a CHILD
                                       process before touching the GPU, relays rank 0's JSON line and exits with its code)

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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_atoms: int):
    """Times the CPU oracle (dense reference formulation, float32) on the host's cores, SURVEY.md section 8d.  Everything in the
    result is MEASURED IN THIS RUN: B = 1 reaction at 16, 32, 64, ... torch threads (doubling up to the host's core count, stopping
    at the first thread count that is slower than the one before - the eager formulation stops scaling early), then B = 8 at the
    best of those; >= 3 timed calls each after one warm-up.  `value` = the best reaction-steps/s of all runs, `cores` = the threads
    that run used; every run is listed with its thread count.  (The round-3 all-core measurement of this function on the MI355X
    host - 256 threads: 84 s per B = 1 call - is on file in profiles/round3_bench_line_cpu_all_cores.json; it is not quoted here.)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import leftnet_oracle as oracle
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    from oareactdiff_amd.synthetic import make_inputs, make_topology
    host_cores = os.cpu_count() or 1
    cfg = dict(PRODUCTION_LEFTNET_CONFIG)
    sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)
    budget = float(os.environ.get("OARD_CPU_BASELINE_SECONDS", "4"))
    runs = []

    def case(B):
        cm, nfs, ei, masks = make_topology(B, n_atoms)
        xh = make_inputs(B, n_atoms, masks, 7, "cpu")
        return lambda: oracle.dynamics_forward(sd, cfg, xh, ei, torch.full((B, 1), 0.5), torch.zeros(B, 1), nfs, cm, 1,
                                               nodeframe="literal")

    def timed(B, threads, call, min_calls, seconds, max_seconds):
        torch.set_num_threads(threads)
        with torch.no_grad():
            call()                                             # warm-up
            calls, t0 = 0, time.perf_counter()
            while (calls < min_calls and time.perf_counter() - t0 < max_seconds) or (time.perf_counter() - t0 < seconds and calls < 40):
                call()
                calls += 1
        dt = time.perf_counter() - t0
        runs.append({"batch": B, "threads": threads, "calls": calls, "seconds": round(dt, 2), "s_per_call": dt / calls,
                     "reaction_steps_per_s": B * calls / dt})
        return runs[-1]

    call1 = case(1)
    th, prev, stopped = min(16, host_cores), None, "the host's core count was reached"
    while True:
        r = timed(1, th, call1, 3, budget, 30.0)
        if prev is not None and r["reaction_steps_per_s"] < prev["reaction_steps_per_s"]:
            stopped = f"{th} threads were slower than {prev['threads']}"
            break
        prev = r
        if th >= host_cores:
            break
        th = min(2 * th, host_cores)
    best1 = max((r for r in runs if r["batch"] == 1), key=lambda r: r["reaction_steps_per_s"])
    timed(8, best1["threads"], case(8), 3, budget, 45.0)
    best = max(runs, key=lambda r: r["reaction_steps_per_s"])
    swept = ", ".join(str(r["threads"]) for r in runs if r["batch"] == 1)
    return {"value": best["reaction_steps_per_s"], "unit": "reaction-steps/s", "cores": best["threads"], "kind": "port",
            "host_cores": host_cores, "cpu_model": cpu_model(), "runs": runs,
            "sample": f"oracle/leftnet_oracle.py (dense reference formulation), float32, {n_atoms}-atom triples; B = 1 at {swept} torch "
                      f"threads on the {host_cores}-core host (doubling; stopped because {stopped}), then B = 8 at {best1['threads']} "
                      f"threads; >= 3 calls / ~{budget:.0f} s each after 1 warm-up, all measured in this run; value = best of all runs "
                      f"(B={best['batch']}, {best['threads']} threads).  All-core record of round 3: "
                      f"profiles/round3_bench_line_cpu_all_cores.json"}


def source_stamp() -> str:
    """Content hash of the kernel sources: profiles carry it, so a traffic figure is only reported for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "oareactdiff_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


PROFILE_TAG = os.environ.get("OARD_PROFILE_TAG", "round6")


def pmc_traffic(kernel: str, prefix: str = None, tag: str = None):
    """HBM bytes per denoising step of all launches of `kernel`, from the committed PMC passes of this round
    (profiles/<prefix>_pmc_{fetch,write}.txt, collected by tools/profile*.sh at the workload the prefix names): FETCH_SIZE
    (KiB; x2 - it under-reports wide streaming reads by 2x on gfx950, MI355X_MICROARCH.md HBM section) + WRITE_SIZE
    (KiB).  Returns (bytes | None, note): None when the profile was taken on other kernel sources than the ones running now
    (profiles/<tag>_source_stamp.txt vs source_stamp()) or is missing."""
    import re
    tag = tag or PROFILE_TAG
    prefix = prefix or tag
    stamp_file = os.path.join(ROOT, "profiles", f"{tag}_source_stamp.txt")
    if not os.path.exists(stamp_file):
        return None, f"profiles/{tag}_source_stamp.txt missing: no PMC pass for this round yet"
    if open(stamp_file).read().strip() != source_stamp():
        return None, f"kernel sources changed since profiles/{prefix}_pmc_*.txt were collected: traffic not reported"
    vals = {}
    for key, fn in (("FETCH_SIZE", f"{prefix}_pmc_fetch.txt"), ("WRITE_SIZE", f"{prefix}_pmc_write.txt")):
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            return None, f"profiles/{fn} missing"
        cur = None
        for line in open(path):
            if line.startswith("=="):
                cur = line
            m = re.search(key + r"\s+per forward = ([0-9.e+]+)", line)
            if m and cur and kernel in cur:
                vals[key] = vals.get(key, 0.0) + float(m.group(1))
    if len(vals) != 2:
        return None, "kernel not found in the PMC summaries"
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        f"HBM bytes per step over all launches of this kernel: 2 x FETCH_SIZE + WRITE_SIZE, profiles/{prefix}_pmc_*.txt (same sources: stamp {source_stamp()})"


def make_training_batch(B: int, n_atoms: int, seed: int, dev):
    """Synthetic Transition1x-shaped batch in the layout of dataset/base_dataset.py:55-88:
    per object {size [B], pos [n,3], one_hot [n,5] int64, charge [n,1] int64, mask [n]} + conditions [B,1], built on the host like a
    collate function does and moved to the device by DDPMTrainer.to_device (which keeps the host copies of mask / size beside the
    device tensors, so a new batch layout costs the step no device -> host copy)."""
    from oareactdiff_amd.trainer import DDPMTrainer
    g = torch.Generator().manual_seed(seed)
    reps = []
    size = torch.full((B,), n_atoms, dtype=torch.long)
    mask = torch.repeat_interleave(torch.arange(B), size)
    n = B * n_atoms
    for _ in range(3):
        pos = torch.randn(n, 3, generator=g)
        pos = pos - (torch.zeros(B, 3).index_add_(0, mask, pos) / n_atoms)[mask]
        typ = torch.multinomial(torch.tensor([0.5, 0.3, 0.1, 0.1]), n, replacement=True, generator=g)
        one_hot = torch.zeros(n, 5, dtype=torch.long)
        one_hot[torch.arange(n), typ] = 1
        charge = torch.tensor([1, 6, 7, 8])[typ].view(n, 1)
        reps.append({"size": size.clone(), "pos": pos, "one_hot": one_hot, "charge": charge, "mask": mask.clone()})
    return DDPMTrainer.to_device((reps, torch.zeros(B, 1)), dev, non_blocking=False)


def edge_counts(B, nf):
    return B * 3 * nf * (3 * nf - 1), B * 3 * nf * (nf - 1)


def fwd_flops(E, A, A_act=None):
    """Algorithmic FLOPs per step of the two hot forward kernels (inter-object edges: no S1 in the first layer - constant
    initial state - and no S3 in the last - nothing reads the result).  EquiMessage is charged for the inner edges INSIDE the cutoff
    only (A_act; SURVEY.md 8d F_req: it is exactly zero on the others, and the kernel skips them)."""
    return {"gcl_edge": 2.0 * (L * MAC_GCL_EDGE * E - (MAC_GCL_S1 + MAC_GCL_S3) * (E - A)),
            "equi_edge": 2.0 * L * MAC_EQUI_EDGE * (A if A_act is None or A_act < 0 else A_act)}


def train_leg(dyn, B, nf, dev, dist, world, steps, warmup, timing=True):
    """One training step = DDPMTrainer.training_step (loss, HIP backward, one all-reduce, adaptive clip, AdamW)."""
    from oareactdiff_amd import _capi
    from oareactdiff_amd.trainer import DDPMTrainer
    # host_sync=False: the clip / skip decision and AdamW's scalars are taken on the device (oard_adamw_step_dev), so the timed steps
    # contain no device -> host read; the returned info is fetched after the timed region.  OARD_BENCH_HOST_SYNC=1: the host decides.
    tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True,
                     host_sync=bool(os.environ.get("OARD_BENCH_HOST_SYNC")), microbatches=int(os.environ.get("OARD_BENCH_MICROBATCHES", "1")))
    # every step a batch the trainer has never seen (new tensors: its layout caches miss, a topology is built and dropped per step), as in
    # a real run over a dataset; OARD_BENCH_CACHED_BATCHES=1: two alternating batches whose layouts stay cached (rounds 2-3 measured that)
    n_b = 2 if os.environ.get("OARD_BENCH_CACHED_BATCHES") else warmup + steps
    rank = dist.get_rank() if dist is not None else 0      # data parallel: every rank trains on ITS shard of the global batch
    batches = [make_training_batch(B, nf, 4321 + k + 100003 * rank, dev) for k in range(n_b)]
    dyn.nan_check = "async"
    for i in range(warmup):
        info = tr.training_step(batches[i % n_b])
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    tr.time_collectives = tr.collectives                    # two event records per step around the ONE all-reduce (no host wait)
    t0 = time.perf_counter()
    for i in range(steps):
        info = tr.training_step(batches[(warmup + i) % n_b])
    torch.cuda.synchronize(dev)
    busy = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    coll = tr.collective_report() if tr.collectives else None
    tr.time_collectives = False
    out = {"ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "batch_per_gpu": B, "loss": info["loss"],
           "host_sync_in_step": bool(tr.host_sync), "new_batch_layout_every_step": n_b > 2,
           "grad_norm": info.get("grad_norm"), "trainable_parameters": int(tr.flat_grad.numel()),
           "all_reduce_bytes": int(tr.flat_grad.numel() * 4) if world > 1 else 0, "seconds": dt, "busy_seconds": busy, "collective": coll}
    # two more steps with per-family kernel timing on rank 0; EVERY rank runs them (each step holds a collective).  The timed
    # steps above run the reverse sweep on two streams (weight-gradient work beside the cotangent chain), where per-launch
    # event times overlap and do not add up; the family pass runs the sweep on ONE stream (debug option train_dual = 0).
    L_ = _capi.lib()
    if timing:
        L_.oard_debug_option(b"train_dual", 0)
        L_.oard_timing_reset()
        L_.oard_timing_enable(1)
    for i in range(2):
        tr.training_step(batches[i % 2])
    torch.cuda.synchronize(dev)
    if timing:
        L_.oard_timing_enable(0)
        L_.oard_debug_option(b"train_dual", int(os.environ.get("OARD_TRAIN_DUAL", "1")))
        E, A = edge_counts(B, nf)
        fam = {}
        for f in ("gcl_edge", "equi_edge", "node", "init", "other", "gcl_edge_bwd", "equi_edge_bwd", "wgrad"):
            ms, n = _capi.timing_get(f)
            fam[f] = {"ms_per_step": ms / 2, "launches_per_step": n / 2}
        ff = fwd_flops(E, A)
        # backward: the dx kernels execute the forward's MACs transposed (last-layer inter-object rows: no S3^T);
        # the weight-gradient GEMMs the same MACs once more (layer-0 inter-object rows: W1c via an outer product)
        fl = {"gcl_edge": ff["gcl_edge"], "equi_edge": ff["equi_edge"],
              "gcl_edge_bwd": 2.0 * (L * MAC_GCL_EDGE * E - MAC_GCL_S3 * (E - A)),
              "equi_edge_bwd": 2.0 * L * (3 * H * W + 9 * H * H) * A,
              "wgrad": 2.0 * (L * (MAC_GCL_EDGE - H) * E - (MAC_GCL_S1 + MAC_GCL_S3) * (E - A)) + 2.0 * L * (3 * H * W + 9 * H * H) * A}
        out["families_ms_per_step"] = {k: round(v["ms_per_step"], 3) for k, v in fam.items()}
        out["families_note"] = "kernel times of a single-stream sweep (additive); ms_per_step is the two-stream step"
        out["hip_kernel_ms_per_step"] = round(sum(v["ms_per_step"] for v in fam.values()), 3)
        out["tflops_by_family"] = {k: round(fl[k] / (fam[k]["ms_per_step"] * 1e-3) / 1e12, 1) for k in fl if fam[k]["ms_per_step"] > 0}
        out["algorithmic_flops_per_step"] = sum(fl.values())
        out["tflops_whole_step"] = sum(fl.values()) / (out["ms_per_step"] * 1e-3) / 1e12
        out["frac_of_f32_mfma_peak_whole_step"] = out["tflops_whole_step"] * 1e12 / PEAK_F32_MFMA
    return out, dt


def new_dynamics(dev, precision="f32"):
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
    cfg = dict(PRODUCTION_LEFTNET_CONFIG)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg), strict=True)
    dyn.nan_check = "async"                 # no host sync inside the step
    if precision == "bf16x3":
        dyn.edge_precision = "bf16x3"       # the split-precision edge kernels for this module's inference calls
        dyn.train_edge_precision = "bf16x3" # ... and for its training-mode forward (--mode train; the backward stays fp32)
    return dyn


class Workload:
    """Fixed-distribution inputs of one (B, atoms) shape, resident in HBM, cycled per step (fresh noise each step)."""

    def __init__(self, B, nf, dev, seed, pos_scale=1.0, n_sets=4):
        from oareactdiff_amd.synthetic import make_inputs, make_topology
        self.B, self.nf, self.dev = B, nf, dev
        cm, nfs, ei, masks = make_topology(B, nf)
        self.cm, self.nfs, self.ei = cm.to(dev), nfs.to(dev), ei.to(dev)
        self.inputs = [make_inputs(B, nf, masks, seed + k, dev, pos_scale) for k in range(n_sets)]
        self.cond = torch.zeros(B, 1, device=dev)
        self.ts = [torch.full((B, 1), (1000 - s) / 1000, device=dev) for s in range(8)]

    def step(self, dyn, i):
        with torch.no_grad():
            dyn(self.inputs[i % len(self.inputs)], self.ei, self.ts[i % len(self.ts)], self.cond, self.nfs, self.cm)


def input_seed(rank: int) -> int:
    """Seed of rank `rank`'s synthetic workload (distinct per rank: a weak-scaling job times N different batches)."""
    return 1234 + 17 * rank


def device_identity(dev):
    """What tells two GPUs apart in the record: index, name, PCI location (domain:bus:device), UUID when torch exposes them."""
    if dev is None or not torch.cuda.is_available():
        return {"device_index": None, "device_name": None, "pci_bus_id": None, "uuid": None}
    p = torch.cuda.get_device_properties(dev)
    pci = None
    if all(hasattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    return {"device_index": dev.index, "device_name": p.name, "pci_bus_id": pci, "uuid": str(getattr(p, "uuid", "")) or None}


def ranks_report(dist, rank, world, dev, dt_local, steps, extra=None, busy=None):
    """The N > 1 line must be verifiable from the record alone: every rank contributes its own wall clock, device identity, host and
    pid (all_gather through the SAME process group the timed region's barriers used), rank 0 prints them with the world size AS THE
    BACKEND REPORTS IT.  A job whose ranks landed on one GPU, or a launcher that started fewer ranks than --gpus, shows up here."""
    import socket
    # ms_per_step: this rank's clock over the whole timed region (its K steps + the closing barrier - what the MAX is taken of);
    # busy_ms_per_step: its own K steps only (device synchronised, before the barrier): the spread between ranks is visible here
    # bound_device_index: the device this rank's LOCAL_RANK selects (torch.cuda.set_device(local_rank) in main(); a dry run has no device
    # and reports the binding it would make); input_seed: the seed of this rank's synthetic inputs - ranks must not time identical data
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    me = dict(device_identity(dev), rank=rank, ms_per_step=dt_local / steps * 1e3,
              busy_ms_per_step=None if busy is None else busy / steps * 1e3, host=socket.gethostname(), pid=os.getpid(),
              local_rank=local_rank, bound_device_index=local_rank if dev is None else dev.index, input_seed=input_seed(rank))
    if extra:
        me.update(extra)
    if dist is None:
        return {"backend": None, "world_size": 1, "per_rank": [me], "distinct_devices": 1}
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, me)
    if rank != 0:
        return None
    ident = {(r["host"], r["pci_bus_id"] or r["uuid"] or r["device_index"]) for r in every}
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "expected_world_size": world,
            "per_rank": sorted(every, key=lambda r: r["rank"]), "distinct_devices": len(ident),
            "slowest_rank": max(every, key=lambda r: r["busy_ms_per_step"] or r["ms_per_step"])["rank"]}


def timed_steps(step, steps, warmup, dev, dist=None):
    """W untimed steps, then exactly K steps between (barrier +) device synchronisations; wall clock of THIS rank."""
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(dev)
    timed_steps.busy = time.perf_counter() - t0             # this rank's own K steps, before it waits for the others
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0


def kernel_families(dyn, step, dev, calls=3):
    """Per-family kernel time per step, HIP events on the launch stream (oard_timing_*), the batch as ONE sub-batch (whole-batch
    launches, nothing overlapping): the figure a roofline fraction is about.  The timed steps run the default schedule (concurrent
    sub-batches), which is faster than the sum of the isolated kernels."""
    from oareactdiff_amd import _capi
    L_ = _capi.lib()
    L_.oard_debug_option(b"parts", 1)
    dyn._topo_cache.clear()
    step(0)
    torch.cuda.synchronize(dev)
    L_.oard_timing_reset()
    L_.oard_timing_enable(1)
    for i in range(calls):
        step(i)
    torch.cuda.synchronize(dev)
    L_.oard_timing_enable(0)
    L_.oard_debug_option(b"parts", int(os.environ.get("OARD_PARTS", "0")))
    dyn._topo_cache.clear()
    fam = {}
    for f in ("gcl_edge", "equi_edge", "node", "init", "other"):
        ms, n = _capi.timing_get(f)
        fam[f] = {"avg_ms": ms / max(n, 1), "launches_per_step": n / calls, "ms_per_step": ms / calls}
    return fam


PEAK_BF16_MFMA = 2.5e15                       # MI355X_MICROARCH.md:42 (dense; the sparse figure is twice that)


def roofline_of(fam, E, A, precision, traffic_prefix=None, A_act=None):
    """`roofline` object of the dominant edge kernel: ALGORITHMIC FLOPs per second in both precisions.  fp32: against the fp32 MFMA
    peak.  bf16x3: both edge families run on bf16 MFMAs - six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block of the fp32 product -
    so the roof is the dense bf16 MFMA peak / 6; the executed bf16 rate is a separate key."""
    flops = fwd_flops(E, A, A_act)
    dom = max(flops, key=lambda f: fam[f]["ms_per_step"])
    oth = [f for f in flops if f != dom][0]
    ach = flops[dom] / (fam[dom]["ms_per_step"] * 1e-3)
    traffic, traffic_note = pmc_traffic("k_" + dom, traffic_prefix)
    roof = {"bound": "mfma", "kernel": "k_" + dom, "achieved": ach / 1e12, "peak": PEAK_F32_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA, "traffic": traffic, "traffic_note": traffic_note,
            "algorithmic_flops_per_step": flops[dom], "launches_per_step": fam[dom]["launches_per_step"],
            "kernel_ms_per_step": fam[dom]["ms_per_step"], "avg_launch_ms": fam[dom]["avg_ms"],
            "families_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in fam.items()},
            "other_kernel": {"kernel": "k_" + oth, "achieved": flops[oth] / (fam[oth]["ms_per_step"] * 1e-3) / 1e12,
                             "frac": flops[oth] / (fam[oth]["ms_per_step"] * 1e-3) / PEAK_F32_MFMA,
                             "algorithmic_flops_per_step": flops[oth], "kernel_ms_per_step": fam[oth]["ms_per_step"]}}
    roof["inner_edges"] = A
    roof["active_inner_edges"] = A if A_act is None or A_act < 0 else A_act      # same-object edges inside the cutoff in the timed inputs
    if traffic is not None:
        roof["hbm_gbps_while_running"] = traffic / (fam[dom]["ms_per_step"] * 1e-3) / 1e9
        t2, _ = pmc_traffic("k_" + oth, traffic_prefix)
        if t2 is not None:
            roof["other_kernel"]["traffic"] = t2
            roof["other_kernel"]["hbm_gbps_while_running"] = t2 / (fam[oth]["ms_per_step"] * 1e-3) / 1e9
    if precision == "bf16x3":
        # `achieved` stays ALGORITHMIC (fp32-equivalent FLOPs, the unit of the fp32 line); the roof it is held against is what the bf16
        # pipe can deliver for this formulation: six v_mfma_f32_16x16x32_bf16 products per fp32 product -> dense bf16 peak / 6.  The
        # executed bf16 rate (6 x, padding not counted) is reported beside it, not in `achieved`.
        peak = PEAK_BF16_MFMA / 6.0
        roof.update({"peak": peak / 1e12, "frac": ach / peak, "executed_bf16_tflops": 6 * ach / 1e12,
                     "mfma_pipe_frac": 6 * ach / PEAK_BF16_MFMA,
                     "precision_note": "achieved = algorithmic fp32 FLOPs per second (comparable with the fp32 line); peak = dense bf16 MFMA "
                                       "peak (2.5 PFLOP/s) / 6 bf16 products per fp32 product; executed_bf16_tflops = 6 x achieved"})
        roof["other_kernel"]["executed_bf16_tflops"] = 6 * roof["other_kernel"]["achieved"]
        roof["other_kernel"]["frac"] = roof["other_kernel"]["achieved"] * 1e12 / peak
    return roof


def sampler_run(dyn, wl, T_run, dev):
    """A genuine ancestral sampling run of T steps (T + 1 network calls + fused sampler kernel + RNG), timed end to end."""
    from oareactdiff_amd.sampler import DiffusionSampler
    B, nf = wl.B, wl.nf
    frag = [torch.full((B,), nf, dtype=torch.long) for _ in range(3)]
    h0 = [x[:, 3:].clone() for x in wl.inputs[0]]
    # Untrained weights predict eps ~ 0 and the ancestral sampler then inflates its state by 1 / alpha_t|s per step: within a few calls
    # every pair is beyond the 10 A cutoff and the run would be timed on an empty radius graph (rounds 1-5 reported exactly that).
    # `gaussian_prior_std=1` adds the ideal denoiser of N(0, 1) data to the network's prediction (one axpy per object and step): the
    # chain keeps the N(0, 1) marginal of the headline's inputs, every call pays for the whole graph - what a trained model pays.
    warm = DiffusionSampler(dyn, "polynomial_2", 4, 1e-5, pos_only=True, gaussian_prior_std=1.0)
    warm.sample(B, frag, conditions=wl.cond, h0=h0)                     # warm-up (topology, buffers)
    smp = DiffusionSampler(dyn, "polynomial_2", T_run, 1e-5, pos_only=True, gaussian_prior_std=1.0)
    # head and tail of the SAME run: events on the loop's stream after network calls 1, 101, T - 100 and T (no host wait inside the loop).
    marks = {}
    want = (1, 101, T_run - 100, T_run) if T_run >= 300 else ()

    def cb(i):
        if i in want:
            marks[i] = torch.cuda.Event(enable_timing=True)
            marks[i].record()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    smp.sample(B, frag, conditions=wl.cond, h0=h0, step_callback=cb)
    torch.cuda.synchronize(dev)
    dts = time.perf_counter() - t1
    per_call = dts / (T_run + 1)
    tail_active = dyn.active_inner_edges()                 # of the LAST network call of the run
    head = None
    if len(marks) == 4:
        head = {"ms_per_call_first_100": marks[1].elapsed_time(marks[101]) / 100.0,
                "ms_per_call_last_100": marks[T_run - 100].elapsed_time(marks[T_run]) / 100.0}
    return {"T": T_run, "network_calls": T_run + 1, "seconds": dts, "batch": B,
            "ms_per_network_call_incl_sampler_step": per_call * 1e3,
            "inner_edges": edge_counts(B, nf)[1], "active_inner_edges_in_the_last_call": tail_active, "head_and_tail": head,
            "reactions_per_sec_measured" if T_run == 1000 else "reactions_per_sec_T1000_projected":
                B / dts if T_run == 1000 else B / (1001 * per_call),
            "final_position_std": float(torch.cat([x[:, :3] for x in smp.last_x]).std()),
            "note": "synthetic (untrained) weights predict eps ~ 0, which makes the ancestral sampler inflate its state beyond the 10 A "
                    "cutoff within a few calls; this run adds the ideal denoiser of N(0, 1) data to the network's prediction "
                    "(DiffusionSampler(gaussian_prior_std=1): one axpy per object and step), so that the chain keeps the N(0, 1) marginal of "
                    "the headline's inputs and every one of the T + 1 network calls runs the whole radius graph "
                    "(active_inner_edges_in_the_last_call == inner_edges), as under a trained model"}


def second_line(dev, B, nf, steps, warmup, quick):
    """The split-precision line (csrc/oard_edge_b3.h) measured by the SAME run as the headline: same B / atoms /
    steps / warm-up, its own module (`edge_precision = "bf16x3"`), outside the headline's timed region.  fp32 stays the headline."""
    dyn = new_dynamics(dev, "bf16x3")
    wl = Workload(B, nf, dev, 1234)
    step = lambda i: wl.step(dyn, i)                      # noqa: E731
    dt = timed_steps(step, steps, warmup, dev)
    assert int(dyn.last_status[0].item()) == 0, "NaN in the second line's timed region"
    E, A = edge_counts(B, nf)
    out = {"precision": "bf16x3", "dtype": "f32 via bf16x3 (the two edge stages: 3 bf16 terms per value, 6 bf16 MFMAs per K block, fp32 "
                                           "accumulate; everything else f32)",
           "value": B * steps / dt, "unit": "reaction-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
           "reactions_per_sec_T1000": B * steps / dt / 1001.0,
           "roofline": roofline_of(kernel_families(dyn, step, dev), E, A, "bf16x3", PROFILE_TAG + "_bf16x3", dyn.active_inner_edges())}
    if not quick:
        out["sampler_loop"] = sampler_run(dyn, wl, 1000, dev)
    return out


def config5(dev, steps=6, warmup=2):
    """BASELINE configs[4] - "Large synthetic fragments (128 atoms/object, dense radius graph) - LDS-tile / HBM-roofline stress": B = 4
    reactions x 3 x 128 atoms (N = 1 536, E = 588 288 edge rows of 2 752 B, A = 195 072 inner edges), fp32, two position scales:
    N(0, 1) (every intra-object pair inside the 10 A cutoff) and x 3 (the cutoff bites: ragged active set, node labelling).  Per
    variant: ms per step (default schedule), per-family kernel time (one sub-batch), TFLOP/s of the two edge kernels against the
    fp32 MFMA peak (EquiMessage runs - and is charged for - the inner edges inside the cutoff only: `active_inner_edges`), HBM GB/s from the
    committed PMC passes of this workload (profiles/<round>_cfg5{n,x3}_pmc_*, tools/profile_cfg5.sh)."""
    B, nf = 4, 128
    E, A = edge_counts(B, nf)
    out = {"workload": f"EGNNDynamics.forward (LEFTNet H=196 R=96 L=6), B={B} reactions x 3 objects x {nf} atoms (N={B * 3 * nf}, E={E}, "
                       f"inner edges A={A}), fp32", "steps": steps, "warmup": warmup, "variants": {}}
    for key, scale in (("n01", 1.0), ("x3", 3.0)):
        dyn = new_dynamics(dev)
        wl = Workload(B, nf, dev, 4242, pos_scale=scale, n_sets=2)
        step = lambda i: wl.step(dyn, i)                  # noqa: E731
        dt = timed_steps(step, steps, warmup, dev)
        assert int(dyn.last_status[0].item()) == 0, "NaN in config 5"
        fam = kernel_families(dyn, step, dev, calls=2)
        roof = roofline_of(fam, E, A, "f32", PROFILE_TAG + "_cfg5" + ("n" if key == "n01" else "x3"), dyn.active_inner_edges())
        out["variants"][key] = {"pos_scale": scale, "ms_per_step": dt / steps * 1e3, "reaction_steps_per_sec": B * steps / dt,
                                "edge_rows_per_sec": E * steps / dt, "roofline": roof}
        del dyn, wl
        torch.cuda.empty_cache()
    return out


def general_edge_lists(dev, B, nf, calls=3):
    """The general-edge-list path (csrc/oard_general.h; DESIGN.md section 13: any `edge_index` that is not the complete graph per sample)
    on the headline batch: (a) the complete graph sent down that path on purpose, beside the production kernels on the same inputs - time per
    call and agreement of the two independent implementations; (b) an `edge_cutoff` graph of the same atoms
    (utils/_graph_tools.py:31-33), which takes that path by itself."""
    from oareactdiff_amd.graph_tools import get_edges_index
    wl = Workload(B, nf, dev, 777, pos_scale=2.0, n_sets=1)
    xh = wl.inputs[0]

    def run(dyn, ei):
        with torch.no_grad():
            out, _ = dyn(xh, ei, wl.ts[0], wl.cond, wl.nfs, wl.cm)           # (first call: tables, workspace)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(calls):
                out, _ = dyn(xh, ei, wl.ts[0], wl.cond, wl.nfs, wl.cm)
            torch.cuda.synchronize(dev)
        return torch.cat([o.reshape(-1) for o in out]).double(), (time.perf_counter() - t0) / calls * 1e3

    prod = new_dynamics(dev)
    a, ms_prod = run(prod, wl.ei)
    gen = new_dynamics(dev)
    gen.edge_list_path = "general"
    b, ms_gen = run(gen, wl.ei)
    assert gen._last_topo.graph is not None and prod._last_topo.graph is None
    pos = torch.cat([x[:, :3] for x in xh])
    cut = get_edges_index(wl.cm, pos=pos, edge_cutoff=4.0, remove_self_edge=True)
    auto = new_dynamics(dev)
    c, ms_cut = run(auto, cut)
    assert auto._last_topo.graph is not None, "an incomplete edge list must take the general path"
    assert bool(torch.isfinite(c).all())
    return {"workload": f"B={B} x 3 x {nf} atoms, positions ~ 2 N(0,1); {calls} calls each", "dtype": "f32 storage, f64 accumulation (float64 MFMA)",
            "complete_graph": {"edges": int(wl.ei.shape[1]), "ms_per_call_general": ms_gen, "ms_per_call_production": ms_prod,
                               "max_abs_difference_over_max_abs": float((a - b).abs().max() / a.abs().max())},
            "edge_cutoff_4A": {"edges": int(cut.shape[1]), "ms_per_call_general": ms_cut}}


def dry_run(args, rank, world, dist, backend):
    """The multi-process skeleton of the timed region without a GPU (OARD_BENCH_DRY=1, tests/test_bench_multirank.py):
    per-rank stand-in steps, barrier on both sides, MAX of the wall clock over the ranks, ONE JSON line on rank 0."""
    from oareactdiff_amd.shard import max_over_ranks
    B = args.batch

    train = args.mode == "train"
    bucket = torch.full((1024,), float(rank + 1)) if train else None

    def step(i):
        time.sleep(0.002 * (rank + 1))                        # ranks differ on purpose: the MAX must pick the slowest
        if train and dist is not None:                        # the training step holds ONE collective: every rank must reach it
            bucket.fill_(float(rank + 1))
            dist.all_reduce(bucket)
            assert float(bucket[0]) == world * (world + 1) / 2
    for i in range(args.warmup):
        step(i)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    busy = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, dist)
    extra = None
    if train and dist is not None:                            # the collective's own time (host clock here; HIP events on the GPU path)
        t1 = time.perf_counter()
        for _ in range(3):
            dist.all_reduce(bucket)
        extra = {"all_reduce_ms_mean": (time.perf_counter() - t1) / 3 * 1e3, "all_reduce_bytes": int(bucket.numel() * 4)}
    ranks = ranks_report(dist, rank, world, None, dt_local, args.steps, extra, busy)
    if rank == 0:
        print(json.dumps({"ranks": ranks, "metric": "training_steps_per_sec" if train else "denoising_steps_per_sec",
                          "value": world * B * args.steps / dt, "unit": "reaction-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "dry run (no GPU work)", "batch_per_gpu": B,
                                     "parallelism": f"dp{world} (one all-reduce per step)" if train
                                     else f"replica x{world} (no collective)"}, "dry_run": True,
                          "backend": backend}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run as a child process.
    Called before this process has made any HIP call (it never will: the child does the work), so no GPU-initialised
    process is ever replaced or forked."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)), file=sys.stderr)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=dict(os.environ, OARD_BENCH_SELF_LAUNCHED="1"))
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])                                       # rank 0's single JSON line
    sys.stdout.flush()
    return proc.returncode if proc.returncode != 0 or lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="reactions per GPU")
    ap.add_argument("--atoms", type=int, default=23, help="atoms per object")
    ap.add_argument("--pos-scale", type=float, default=1.0, help="scale of the N(0,1) positions (3: the 10 A cutoff bites; BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("sample", "train"), default="sample",
                    help="sample: one denoising call per step (BASELINE configs 2/3); train: one training step (config 4)")
    ap.add_argument("--quick", action="store_true", help="skip the full T=1000 sampling run and the training leg")
    ap.add_argument("--precision", choices=("f32", "bf16x3"), default="f32",
                    help="bf16x3: the SECOND line - the two edge stages on the split-precision kernels (three bf16 terms per fp32 value, six "
                         "bf16 MFMAs per K block, fp32 accumulation: fp32-grade results, csrc/oard_edge_b3.h); inference only; the "
                         "fp32 kernels stay the default and the headline")
    ap.add_argument("--no-graph", action="store_true", help="launch-bound batches (B <= 8): eager launches instead of hipGraph replay")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong job size")
    dist = None
    backend = os.environ.get("OARD_BENCH_BACKEND", "nccl")    # "gloo" only to exercise the N>1 code path on a 1-GPU box
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    dry = bool(os.environ.get("OARD_BENCH_DRY"))              # no GPU work: only the launcher / rank / reduction plumbing (CPU test)
    # OARD_BENCH_FORCE_DIST=1 under the launcher with ONE rank: the process group, the barriers and the MAX-reduce of the N > 1 path
    # run anyway (RCCL smoke test of exactly this code on a one-GPU box, tests/test_rccl_single_rank.py)
    force_dist = bool(os.environ.get("OARD_BENCH_FORCE_DIST")) and "MASTER_PORT" in os.environ
    if force_dist:
        os.environ["OARD_FORCE_COLLECTIVES"] = "1"            # DDPMTrainer: broadcast / bucket all-reduce with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dry:
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if dry:
        return dry_run(args, rank, world, dist, backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from oareactdiff_amd import _capi  # noqa: F401  (loads the library: fails loudly if it is missing)

    B, nf = args.batch, args.atoms
    dyn = new_dynamics(dev, args.precision)
    wl = Workload(B, nf, dev, input_seed(rank), pos_scale=args.pos_scale)

    if args.mode == "train":
        leg, dt = train_leg(dyn, B, nf, dev, dist, world, args.steps, args.warmup, timing=(rank == 0))       # collective inside: all ranks
        dt_local = dt
        if dist is not None:
            tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        ranks = ranks_report(dist, rank, world, dev, dt_local, args.steps, leg.get("collective"), leg.get("busy_seconds"))
        if rank == 0:
            E, A = edge_counts(B, nf)
            out = {"metric": "training_steps_per_sec", "value": world * B * args.steps / dt, "unit": "reaction-steps/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": "f32" if args.precision == "f32" else "f32; the training-mode forward's two edge stages via bf16x3 (3 bf16 terms per value, fp32 accumulate), backward f32",
                   "data": "synthetic",
                   "config": {"workload": f"DDPMTrainer.training_step (loss, HIP backward, gradient all-reduce, adaptive clip, "
                                          f"AdamW amsgrad), LEFTNet H=196 R=96 L=6, B={B} reactions/GPU x 3 objects x {nf} atoms "
                                          f"(N={B * 3 * nf}, E={E}), pos_only training as train_ts1x.py",
                              "batch_per_gpu": B, "atoms_per_object": nf,
                              "parallelism": f"dp{world} (one flat fp32 gradient bucket, one all-reduce per step)",
                              "launch": "host decides the clipping (one device -> host read per step)" if os.environ.get("OARD_BENCH_HOST_SYNC")
                                        else "no host sync inside the step (clip / skip decision and AdamW scalars on the device)"},
                   "train_step": leg, "ranks": ranks}
            if leg.get("tflops_whole_step"):
                out["roofline"] = {"bound": "mfma", "kernel": "whole training step (forward + backward edge kernels' algorithmic FLOPs)",
                                   "achieved": leg["tflops_whole_step"], "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                                   "frac": leg["frac_of_f32_mfma_peak_whole_step"], "traffic": None}
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    def eager_step(i):
        wl.step(dyn, i)

    # Launch-bound batches (B <= 8: ~100 launches of a few microseconds each per call): the same call captured once per
    # input set as a hipGraph and replayed - same kernels, same arithmetic (tests: bit-identical), no host launch cost.
    use_graph = B <= 8 and not args.no_graph
    graphs = []
    if use_graph:
        eager_step(0)                                          # topology, packed weights, workspace, LDS attributes
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            eager_step(0)
        torch.cuda.current_stream(dev).wait_stream(side)
        for k in range(len(wl.inputs)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                eager_step(k)
            graphs.append(g)

    def step(i):
        if graphs:
            graphs[i % len(graphs)].replay()
        else:
            eager_step(i)

    dt = timed_steps(step, args.steps, args.warmup, dev, dist)      # W warm-up steps, barrier + sync, K steps, sync + barrier
    dt_local = dt
    if dist is not None:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ranks = ranks_report(dist, rank, world, dev, dt_local, args.steps, None, timed_steps.busy)
    if not os.environ.get("OARD_BENCH_ALLOW_NAN"):          # (kernel ablation experiments produce garbage on purpose)
        assert int(dyn.last_status[0].item()) == 0, "NaN in the timed region"

    # ---- everything below is outside the timed region, on rank 0 -------------------------------------------------------------------
    E, A = edge_counts(B, nf)
    roof = None
    skip = set(filter(None, os.environ.get("OARD_BENCH_SKIP", "").split(",")))     # debugging: legs to leave out
    std = (B, nf) == (64, 23)
    if rank == 0 and "roofline" not in skip:               # per-kernel durations (HIP events on the launch stream)
        prefix = (PROFILE_TAG if args.precision == "f32" else PROFILE_TAG + "_bf16x3") if std else "no-PMC-pass-for-this-shape"
        roof = roofline_of(kernel_families(dyn, eager_step, dev), E, A, args.precision, prefix, dyn.active_inner_edges())

    # the real sampling loop (row N1): the BASELINE metric's reactions/s, MEASURED (T = 1000 unless --quick)
    sampler_leg = train = second = cfg5 = general = None
    extra = rank == 0 and world == 1 and not os.environ.get("OARD_BENCH_ALLOW_NAN")     # N > 1: the other ranks wait in a barrier meanwhile
    if extra and "sampler" not in skip:
        sampler_leg = sampler_run(dyn, wl, 12 if args.quick else 1000, dev)
    full = extra and not args.quick and args.precision == "f32"
    if full:
        # the remaining legs get what their stand-alone modes have: fresh modules and an allocator that starts empty (the training
        # step walks ~25 GB of tape / scratch; carved out of the other legs' left-overs it ran 5-8 % slower: 83 vs 77 ms measured)
        graphs = wl = None
        del dyn
        torch.cuda.empty_cache()
        if "train" not in skip:
            train, _ = train_leg(new_dynamics(dev), B, nf, dev, None, 1, 10, 3)      # warm-up as in `--mode train`
            torch.cuda.empty_cache()
        if "second_line" not in skip:
            second = second_line(dev, B, nf, args.steps, args.warmup, args.quick)
            torch.cuda.empty_cache()
        if "config5" not in skip:
            cfg5 = config5(dev)
            torch.cuda.empty_cache()
        if "general" not in skip:
            general = general_edge_lists(dev, B, nf)
            torch.cuda.empty_cache()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": "denoising_steps_per_sec", "value": value, "unit": "reaction-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f32 via bf16x3 (the two edge stages: 3 bf16 terms per value, 6 bf16 MFMAs per K block, fp32 accumulate; everything else f32)",
            "data": "synthetic",
            "config": {"workload": f"EGNNDynamics.forward (LEFTNet H=196 R=96 L=6), B={B} reactions/GPU x 3 objects x "
                                   f"{nf} atoms, complete graph per reaction (N={B * 3 * nf}, E={E}), T=1000 sampler step shape",
                       "batch_per_gpu": B, "atoms_per_object": nf, "parallelism": f"replica x{world} (no collective)",
                       "launch": "hipGraph replay (one captured call per input set)" if use_graph else "eager"},
            "batch_steps_per_sec_per_gpu": args.steps / dt,
            "reactions_per_sec_T1000": value / 1001.0,
            "ranks": ranks,
            "roofline": roof,
            "sampler_loop": sampler_leg,
            "train_step": train,
            "second_line": second,
            "config5": cfg5,
            "general_edge_lists": general,
        }
        if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(nf)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
