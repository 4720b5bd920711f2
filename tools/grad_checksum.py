"""Deterministic training steps at the bench shape (B = 64 x 3 x 23 atoms, seeded noise / time steps): prints the loss and a checksum of
the flat gradient after every step.  Two builds of the library that only differ in the ORDER OF WORK (workgroup -> node maps, launch
shapes) must print identical lines; used with OARD_LIB=... for A/B builds.  usage: python tools/grad_checksum.py [steps] [B]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oareactdiff_amd.trainer import DDPMTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
torch.manual_seed(1234)
torch.cuda.manual_seed(1234)
dyn = bench.new_dynamics(dev)
tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
batches = [bench.make_training_batch(B, 23, 4321 + k, dev) for k in range(2)]
for i in range(steps):
    info = tr.training_step(batches[i % 2])
    torch.cuda.synchronize()
    g = tr.flat_grad.detach().cpu()
    print(f"step {i}: loss {info['loss']:.9g} grad_norm {info.get('grad_norm')} |g|_1 {float(g.abs().sum()):.9g} "
          f"sha {hashlib.sha256(g.numpy().tobytes()).hexdigest()[:16]}", flush=True)
