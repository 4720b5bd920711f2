"""Wall-clock (host) time of the pieces of a fused training step on never-seen batches (monkeypatched timers, no profiler)."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oareactdiff_amd import trainer as T, training, dynamics, loss, graph_tools
from oareactdiff_amd.trainer import DDPMTrainer
acc = collections.OrderedDict()
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or name
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
dev = torch.device("cuda:0")
dyn = bench.new_dynamics(dev)
tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True, host_sync=False)
for o, n in ((loss.DiffusionLoss, "_layout"), (dynamics.EGNNDynamics, "_get_train_topology"), (dynamics.EGNNDynamics, "_get_packed"),
             (dynamics.EGNNDynamics, "_get_packed_bwd"), (dynamics.EGNNDynamics, "_train_inputs"), (dynamics.EGNNDynamics, "_run_forward_train"),
             (dynamics.EGNNDynamics, "_ordered_tensors"), (training, "backward_sweep"), (training, "gradient_table"),
             (training.TrainTopology, "__init__", ), (DDPMTrainer, "_fused_forward_backward"), (DDPMTrainer, "_fused_step")):
    wrap(o, n, f"{o.__name__}.{n}")
cached = [bench.make_training_batch(B, 23, 1 + k, dev) for k in range(2)]
fresh = [bench.make_training_batch(B, 23, 100 + k, dev) for k in range(12)]
for i in range(4): tr.training_step(cached[i % 2])
torch.cuda.synchronize()
for label, bs in (("cached", [cached[i % 2] for i in range(12)]), ("fresh", fresh)):
    acc.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in bs: tr.training_step(b)
    host = time.perf_counter() - t0; torch.cuda.synchronize()
    print(label, f"host {host / 12 * 1e3:.2f} ms/step:", "  ".join(f"{k.split('.')[-1]} {v / 12 * 1e3:.2f}" for k, v in acc.items()))
