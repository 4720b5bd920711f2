"""Determinism probe (run on the GPU box): the same inference call / the same training step twice, bit for bit.
usage: python tools/det_probe.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_training_batch, new_dynamics, Workload
from oareactdiff_amd import _capi
from oareactdiff_amd.trainer import DDPMTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
L = _capi.lib()
for kv in os.environ.get("PROBE_OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        assert L.oard_debug_option(k.encode(), int(v)) == 0
for parts in (1, 0):
    L.oard_debug_option(b"parts", parts)
    dyn = new_dynamics(dev)
    wl = Workload(B, 23, dev, 1234)
    outs = []
    for rep in range(4):
        with torch.no_grad():
            o, _ = dyn(wl.inputs[0], wl.ei, wl.ts[0], wl.cond, wl.nfs, wl.cm)
        torch.cuda.synchronize()
        outs.append([x.clone() for x in o])
    same = [all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:]]
    print(f"inference B={B} parts={parts}: repeat runs identical: {same}")
L.oard_debug_option(b"parts", 0)
batch = make_training_batch(B, 23, 100, dev)
res = []
for rep in range(3):
    dyn = new_dynamics(dev)
    tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
    torch.manual_seed(1234)
    info = tr.training_step(batch)
    torch.cuda.synchronize()
    res.append((tr.flat_grad.clone(), info["loss"]))
print("training step: losses", [r[1] for r in res], "grads identical:", [bool(torch.equal(res[0][0], r[0])) for r in res[1:]],
      "max diff", [float((res[0][0] - r[0]).abs().max()) for r in res[1:]])
batches = [make_training_batch(B, 23, 100 + k, dev) for k in range(2)]
res = []
for dual in (1, 1, 0):
    L.oard_debug_option(b"train_dual", dual)
    dyn = new_dynamics(dev)
    tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
    torch.manual_seed(1234)
    grads, losses = [], []
    for i in range(3):
        info = tr.training_step(batches[i % 2])
        torch.cuda.synchronize()
        grads.append(tr.flat_grad.clone()); losses.append(info["loss"])
    res.append((grads, losses, tr.flat_param.clone()))
L.oard_debug_option(b"train_dual", 1)
for k, r in enumerate(res[1:]):
    print("run", k + 1, "dual", (1, 0)[k], "losses equal", r[1] == res[0][1], "per-step grad max diff",
          [float((a - b).abs().max()) for a, b in zip(res[0][0], r[0])], "params equal", bool(torch.equal(res[0][2], r[2])))
