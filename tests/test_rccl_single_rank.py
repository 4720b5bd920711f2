"""The data-parallel training step over RCCL itself (backend "nccl" on ROCm), as far as a ONE-GPU box allows: a one-rank
process group on cuda:0, the trainer's collectives forced on (OARD_FORCE_COLLECTIVES: the initial parameter / buffer broadcast
with its checksum all-reduces, the flat [gradients | non-finite flag] bucket all-reduce of every step, bench.py's MAX-over-ranks
of the wall clock).  A one-rank SUM / broadcast is the identity, so three fused training steps must end in bit-identical weights
with and without the collectives.  The two-rank arithmetic of the same code is covered on CPU by tests/test_trainer_gloo.py
(gloo); what this adds is that the device buffers the trainer hands to torch.distributed are ones RCCL accepts and that the HIP
step and the collectives order correctly on the stream.  Runs in a child process (its own process group, its own environment)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
from bench import make_training_batch
from oareactdiff_amd.dynamics import EGNNDynamics
from oareactdiff_amd.shard import max_over_ranks
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict
from oareactdiff_amd.trainer import DDPMTrainer

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2)
sd = synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg)
batches = [make_training_batch(4, 9, 200 + k, dev) for k in range(2)]

def run(force):
    if force:
        os.environ["OARD_FORCE_COLLECTIVES"] = "1"
    else:
        os.environ.pop("OARD_FORCE_COLLECTIVES", None)
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0, condition_nf=1, device=dev)
    dyn.load_state_dict(sd, strict=True)
    tr = DDPMTrainer(dyn, timesteps=1000, norm_values=(1.0, 4.0, 10.0), scales=(1.0, 2.0, 1.0), pos_only=True)
    assert tr.collectives == force and tr.world == 1
    torch.manual_seed(77)
    infos = [tr.training_step(batches[i % 2]) for i in range(3)]
    assert all(i["skipped"] == 0 for i in infos)
    return tr.flat_param.clone(), [i["loss"] for i in infos]

plain_w, plain_l = run(False)                        # no process group yet: the single-process path
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["OARD_TEST_PORT"], rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
rccl_w, rccl_l = run(True)
assert rccl_l == plain_l, (rccl_l, plain_l)
assert torch.equal(rccl_w, plain_w)
# bench.py's reduction of the timed region, on a device tensor over RCCL (shard.max_over_ranks skips one-rank groups: call the collective)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 1.25 and max_over_ranks(1.25, dist, dev) == 1.25
dist.barrier()
torch.cuda.synchronize(dev)
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK", plain_l)
"""


def test_training_step_collectives_over_rccl_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OARD_TEST_PORT=str(port), MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("OARD_FORCE_COLLECTIVES", None)
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("mode", ["sample", "train"])
def test_bench_launch_line_over_rccl_one_rank(mode):
    """The driver's launch line with ONE rank and OARD_BENCH_FORCE_DIST=1: bench.py initialises the nccl process group with
    device_id, and runs the barriers, the MAX all-reduce of the wall clock on a device tensor, (train) the trainer's broadcast and
    bucket all-reduce, and destroy_process_group - the code of the N > 1 path, over RCCL, on the GPU that is there."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OARD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8",
           "--quick", "--no-cpu-baseline"] + (["--mode", "train"] if mode == "train" else [])
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    # the self-verifying part of the line, over RCCL on real hardware: the backend's own world size, the rank's device identity and clock
    rk = d["ranks"]
    assert rk["backend"] == "nccl" and rk["world_size"] == 1 and rk["distinct_devices"] == 1 and len(rk["per_rank"]) == 1
    me = rk["per_rank"][0]
    assert me["rank"] == 0 and me["device_index"] == 0 and me["device_name"] and me["pid"] > 0
    assert me["pci_bus_id"] or me["uuid"], me                                  # something that tells two GPUs apart
    assert abs(me["ms_per_step"] - d["ms_per_step"]) < 1e-6 and 0 < me["busy_ms_per_step"] <= me["ms_per_step"] + 1e-6
    if mode == "train":                                                      # the bucket all-reduce's own device time and size
        assert me["all_reduce_bytes"] == 4 * (d["train_step"]["trainable_parameters"] + 1) and me["all_reduce_ms_mean"] > 0
        assert me["all_reduce_calls"] == 3
