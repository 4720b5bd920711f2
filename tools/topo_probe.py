import os, sys, time, ctypes as C
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from oareactdiff_amd import training, _capi
from oareactdiff_amd.graph_tools import get_edges_index, get_n_frag_switch
dev = torch.device("cuda:0")
dyn = bench.new_dynamics(dev)
reps, cond = bench.make_training_batch(64, 23, 1, dev)
masks = [r["mask"] for r in reps]; sizes = [r["size"] for r in reps]
def T(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
cm = torch.cat(masks)
ms, ei = T(lambda: get_edges_index(cm, remove_self_edge=True)); print(f"get_edges_index {ms:.2f} ms")
ms, nfs = T(lambda: get_n_frag_switch(sizes)); print(f"get_n_frag_switch {ms:.2f} ms")
cfg = dyn._config() if hasattr(dyn, "_config") else None
print([a for a in dir(dyn) if "cfg" in a.lower() or "config" in a.lower()])
st = torch.cuda.current_stream().cuda_stream
ms, tp = T(lambda: training.TrainTopology(cfg, cm, nfs, st, edge_index=ei)); print(f"TrainTopology with edge_index check {ms:.2f} ms")
ms, tp = T(lambda: training.TrainTopology(cfg, cm, nfs, st)); print(f"TrainTopology without check {ms:.2f} ms")
L = _capi.lib()
def create_only():
    cmc = cm.detach().to("cpu", torch.int64).contiguous(); nfc = nfs.detach().to("cpu", torch.int64).contiguous()
    h = C.c_void_p()
    _capi.check(L.oard_topology_create_parts(C.byref(cfg), C.cast(cmc.data_ptr(), C.POINTER(C.c_int64)), C.cast(nfc.data_ptr(), C.POINTER(C.c_int64)), cmc.numel(), 1, C.byref(h)), "x")
    return h
hs = []
ms, h = T(lambda: hs.append(create_only())); print(f"oard_topology_create_parts (+ mask copies) {ms:.2f} ms")
t0 = time.perf_counter()
for h in hs: L.oard_topology_destroy(h)
print(f"oard_topology_destroy {(time.perf_counter() - t0) / len(hs) * 1e3:.2f} ms")
