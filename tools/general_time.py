"""Time of one call of the general-edge-list path (csrc/oard_general.h) beside the production kernels on the same complete graph, and
their agreement.  usage (GPU box): python tools/general_time.py [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402
from oareactdiff_amd.synthetic import make_inputs, make_topology  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(PRODUCTION_LEFTNET_CONFIG)
sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=42)
for B in [int(x) for x in sys.argv[1:]] or [1, 8, 64]:
    nf = 23
    cm, nfs, ei, masks = make_topology(B, nf)
    xh = [x.to(dev) for x in make_inputs(B, nf, masks, 5, "cpu")]
    args = (xh, ei.to(dev), torch.full((B, 1), 0.5, device=dev), torch.zeros(B, 1, device=dev), nfs.to(dev), cm.to(dev))
    outs = {}
    paths = ("auto", "general") if os.environ.get("OARD_GENERAL_TIME_SKIP_THREADS") else ("auto", "general", "general_threads")
    for path in paths:
        os.environ["OARD_GENERAL_GEMM"] = "threads" if path == "general_threads" else "matrix"     # read by the library per call
        dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
        dyn.load_state_dict(sd, strict=True)
        dyn.edge_list_path = "auto" if path == "auto" else "general"
        dyn.nan_check = "async"
        with torch.no_grad():
            dyn(*args)
            torch.cuda.synchronize()
            n = 3 if path != "auto" else 10
            t0 = time.perf_counter()
            for _ in range(n):
                o, _ = dyn(*args)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        outs[path] = (torch.cat([x.reshape(-1) for x in o]).double().cpu(), dt)
        del dyn
    if "general_threads" not in outs:
        outs["general_threads"] = (outs["general"][0], float("nan"))
    a, b, c = outs["auto"][0], outs["general"][0], outs["general_threads"][0]
    print(f"B = {B} (E = {ei.shape[1]}): production kernels {outs['auto'][1] * 1e3:.2f} ms, general path {outs['general'][1] * 1e3:.1f} ms per call "
          f"(dense layers on plain threads, OARD_GENERAL_GEMM=threads: {outs['general_threads'][1] * 1e3:.1f} ms); "
          f"max |difference| / max |.| = {float((a - b).abs().max() / a.abs().max()):.2e} (threads vs matrix pipe: {float((c - b).abs().max() / a.abs().max()):.2e})")
