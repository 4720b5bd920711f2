"""Is the training step host-bound at a given batch size?  Host time to ENQUEUE a step (no synchronisation inside the loop; the one
host read per step is still there) against the wall time per step.  python tools/host_bound_probe.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oareactdiff_amd.dynamics import EGNNDynamics  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402
from oareactdiff_amd.trainer import DDPMTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
dev = torch.device("cuda:0")
cfg = dict(PRODUCTION_LEFTNET_CONFIG)
dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
dyn.load_state_dict(synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg), strict=True)
dyn.nan_check = "async"
tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
batches = [bench.make_training_batch(B, 23, 4321 + k, dev) for k in range(2)]
for i in range(3):
    tr.training_step(batches[i % 2])
torch.cuda.synchronize(dev)
# (1) wall per step
t0 = time.perf_counter()
for i in range(10):
    tr.training_step(batches[i % 2])
torch.cuda.synchronize(dev)
wall = (time.perf_counter() - t0) / 10
# (2) host time of the forward + backward enqueue only (the part before the step's one host read)
host = []
for i in range(10):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    tr._fused_forward_backward(batches[i % 2])
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize(dev)
print(f"B {B}: wall {wall * 1e3:.2f} ms per step; host time to enqueue loss-prepare + forward + sweep {sum(host) / len(host) * 1e3:.2f} ms")
