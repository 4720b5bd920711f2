"""What a NEVER-SEEN batch layout costs the training step (real training has a new layout every step; bench.py's two alternating
batches hit the layout caches): the time of DDPMTrainer.training_step on fresh batches (new tensors, same shape) against the cached one.
usage: python tools/topo_cost.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oareactdiff_amd.trainer import DDPMTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
dyn = bench.new_dynamics(dev)
tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True, host_sync=False)
cached = [bench.make_training_batch(B, 23, 4321 + k, dev) for k in range(2)]
for i in range(4):
    tr.training_step(cached[i % 2])
torch.cuda.synchronize()


def timed(batches):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        tr.training_step(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(batches) * 1e3


print(f"cached layouts: {timed([cached[i % 2] for i in range(10)]):.2f} ms per step")
fresh = [bench.make_training_batch(B, 23, 9000 + k, dev) for k in range(10)]
print(f"fresh layouts (same sizes, new tensors): {timed(fresh):.2f} ms per step")
# ragged sizes: every batch a different layout
import random
random.seed(0)
rag = []
for k in range(10):
    reps, cond = bench.make_training_batch(B, 23, 9100 + k, dev)
    rag.append((reps, cond))
print(f"again the same 10 (now cached by identity?): {timed(fresh):.2f} ms per step")
