"""Host time of DDPMTrainer.training_step (the call's duration without waiting for the device) on cached and on never-seen batches, with a
cProfile of the never-seen case.  usage: python tools/host_time_probe.py [B]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oareactdiff_amd.trainer import DDPMTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
dev = torch.device("cuda:0")
dyn = bench.new_dynamics(dev)
tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True, host_sync=False)
cached = [bench.make_training_batch(B, 23, 1 + k, dev) for k in range(2)]
fresh = [bench.make_training_batch(B, 23, 100 + k, dev) for k in range(24)]
for i in range(4):
    tr.training_step(cached[i % 2])
torch.cuda.synchronize()


def host(batches):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        tr.training_step(b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / len(batches) * 1e3, (t2 - t0) / len(batches) * 1e3


print("cached: host %.2f ms per step, wall %.2f" % host([cached[i % 2] for i in range(12)]))
print("fresh:  host %.2f ms per step, wall %.2f" % host(fresh[:12]))
pr = cProfile.Profile()
pr.enable()
for b in fresh[12:]:
    tr.training_step(b)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
