"""The five long weight-gradient products of one layer at B = 64 (E = 300 288 rows, A = 97 152 inner rows), each launched `reps` times:
run under rocprofv3 (--kernel-trace --stats, or a --pmc pass) to read per-kernel durations / counters of k_wgrad_t16 and its reduce
passes in isolation.  usage: python tools/wgrad_probe.py [reps] [only-shape-index]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oareactdiff_amd import training  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
dev = torch.device("cuda:0")
E, A = 300288, 97152
shapes = [("edge_out_trans", E, 688, 684, 684, 684, 208, 196, 196, 196, False, True),
          ("edge_mlp.1", E, 208, 196, 196, 196, 208, 196, 196, 196, True, True),
          ("edge_mlp.0", E, 208, 196, 196, 196, 688, 684, 684, 684, False, False),
          ("dir_proj.2", A, 624, 196, 208, 588, 592, 588, 588, 588, True, True),
          ("dir_proj.0", A, 592, 588, 588, 588, 688, 684, 684, 684, False, True)]


class Owner:
    pass


st = torch.cuda.current_stream().cuda_stream
for k, (name, rows, ncY, ol, op, MO, ncX, il, ip, MI, silu, bias) in enumerate(shapes):
    if only >= 0 and k != only:
        continue
    dY, X = torch.randn(rows, ncY, device=dev), torch.randn(rows, ncX, device=dev)
    own = Owner()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    training._wgrad(dY, ncY, ol, op, MO, X, ncX, silu, il, ip, MI, rows, bias, own, st)
    a.record()
    for _ in range(reps):
        training._wgrad(dY, ncY, ol, op, MO, X, ncX, silu, il, ip, MI, rows, bias, own, st)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name:15s} rows {rows}: {ms:.3f} ms per product (kernel + reduce passes) = {2.0 * rows * MO * MI / ms / 1e9:.1f} TF/s algorithmic")
