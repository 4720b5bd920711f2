"""Probe: one B=64 forward loop vs two interleaved B=32 loops on two streams (no join between the halves).
usage: python tools/two_half_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402
from oareactdiff_amd.synthetic import make_inputs, make_topology  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cfg = dict(PRODUCTION_LEFTNET_CONFIG)
sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)


def mk(B, seed):
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    dyn.nan_check = "async"
    cm, nfs, ei, masks = make_topology(B, 23)
    args = (ei.to(dev), torch.full((B, 1), 0.5, device=dev), torch.zeros(B, 1, device=dev), nfs.to(dev), cm.to(dev))
    return dyn, make_inputs(B, 23, masks, seed, dev), args


def loop(items, streams, n):
    for (dyn, xh, args), st in zip(items, streams):
        with torch.cuda.stream(st), torch.no_grad():
            dyn(xh, *args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for (dyn, xh, args), st in zip(items, streams):
            with torch.cuda.stream(st), torch.no_grad():
                dyn(xh, *args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


one = [mk(64, 1)]
print("one B=64 loop          : %.3f ms per 64 reactions" % loop(one, [torch.cuda.current_stream()], steps))
two = [mk(32, 1), mk(32, 2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
print("two B=32 loops, 2 streams: %.3f ms per 64 reactions" % loop(two, s, steps))
four = [mk(16, k) for k in range(4)]
s4 = [torch.cuda.Stream() for _ in range(4)]
print("four B=16 loops, 4 streams: %.3f ms per 64 reactions" % loop(four, s4, steps))
