"""Probe: does running two half-batches concurrently on two streams beat one full batch? (GPU box)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oareactdiff_amd.dynamics import EGNNDynamics
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
from oareactdiff_amd.synthetic import make_inputs, make_topology

dev = torch.device("cuda:0")
cfg = dict(PRODUCTION_LEFTNET_CONFIG)
sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)


def mk(B):
    d = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
    d.load_state_dict(sd); d.nan_check = "async"
    cm, nfs, ei, masks = make_topology(B, 23)
    a = (make_inputs(B, 23, masks, 1, dev), ei.to(dev), torch.full((B, 1), 0.5, device=dev), torch.zeros(B, 1, device=dev), nfs.to(dev), cm.to(dev))
    return d, a


def run(parts, steps=20):
    streams = [torch.cuda.Stream() for _ in parts]
    for _ in range(3):
        for (d, a), s in zip(parts, streams):
            with torch.cuda.stream(s), torch.no_grad():
                d(*a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (d, a), s in zip(parts, streams):
            with torch.cuda.stream(s), torch.no_grad():
                d(*a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("1 x 64 :", round(run([mk(64)]), 3), "ms/step")
print("2 x 32 :", round(run([mk(32), mk(32)]), 3), "ms/step")
print("4 x 16 :", round(run([mk(16) for _ in range(4)]), 3), "ms/step")
print("2 x 64 :", round(run([mk(64), mk(64)]) / 2, 3), "ms per 64-reaction step")
