"""Soak run of the fused training step (run on the GPU box): N steps at batch size B on alternating synthetic batches; prints the
step time per block of 50 steps, the loss, the allocator's footprint at the start and at the end (it must not grow) and the number
of skipped steps.  python tools/train_soak.py [B] [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402
from oareactdiff_amd.trainer import DDPMTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
cfg = dict(PRODUCTION_LEFTNET_CONFIG)
dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
dyn.load_state_dict(synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg), strict=True)
dyn.nan_check = "async"
tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
batches = [bench.make_training_batch(B, 23, 4321 + k, dev) for k in range(4)]
for i in range(3):
    tr.training_step(batches[i % 4])
torch.cuda.synchronize(dev)
m0, r0 = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
losses = []
t0 = time.perf_counter()
for i in range(N):
    losses.append(tr.training_step(batches[i % 4])["loss"])
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        print(f"steps {i - 48:4d}-{i + 1:4d}: {(t1 - t0) / 50 * 1e3:7.2f} ms per step, mean loss {sum(losses[-50:]) / 50:.4f}", flush=True)
        t0 = time.perf_counter()
m1, r1 = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
print(f"allocated {m0 / 2**30:.2f} -> {m1 / 2**30:.2f} GiB, reserved {r0 / 2**30:.2f} -> {r1 / 2**30:.2f} GiB, skipped steps {tr.skipped_steps}, "
      f"finite losses {all(x == x and abs(x) < 1e30 for x in losses)}")
assert m1 <= m0 + (64 << 20) and tr.skipped_steps == 0
