"""Does the denoising step depend on how many OTHER streams the process created before the library's?  (The runtime serves streams from 4
hardware queues.)  usage: python tools/queue_probe.py <dummy streams created and used first> [after: create them after the first forward]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
after = len(sys.argv) > 2
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
dummies = []


def make():
    for _ in range(k):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            torch.zeros(8, device=dev).add_(1)
        dummies.append(s)
    torch.cuda.synchronize()


if not after:
    make()
dyn = bench.new_dynamics(dev)
wl = bench.Workload(64, 23, dev, 1234)
with torch.no_grad():
    for i in range(3):
        wl.step(dyn, i)
    if after:
        make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        wl.step(dyn, i)
    torch.cuda.synchronize()
print(f"{k} dummy streams {'after' if after else 'before'} the library's: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
