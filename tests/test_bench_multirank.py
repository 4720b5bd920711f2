"""bench.py under the driver's multi-GPU launch line, on CPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
--master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` with the gloo backend and OARD_BENCH_DRY=1 (stand-in steps, no
GPU work).  Proves the rank / environment handling, the barriers, the MAX-over-ranks wall clock, that exactly one JSON
line comes out (rank 0) and that every rank exits cleanly.  On the 8-GPU node the same code runs with backend nccl = RCCL."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo_dry_run():
    env = dict(os.environ, OARD_BENCH_BACKEND="gloo", OARD_BENCH_DRY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dry_run"]
    # MAX over ranks: rank 1 sleeps 4 ms per step, rank 0 only 2 ms
    assert d["ms_per_step"] >= 3.9
    assert abs(d["value"] - 2 * 64 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-6     # whole-job aggregate
    # the line verifies itself: world size as the backend reports it, one entry per rank with its own clock / host / pid
    rk = d["ranks"]
    assert rk["backend"] == "gloo" and rk["world_size"] == 2 and rk["expected_world_size"] == 2
    assert [r["rank"] for r in rk["per_rank"]] == [0, 1] and len({r["pid"] for r in rk["per_rank"]}) == 2
    assert rk["per_rank"][1]["busy_ms_per_step"] >= 3.9 > rk["per_rank"][0]["busy_ms_per_step"] >= 1.9 and rk["slowest_rank"] == 1
    assert abs(max(r["ms_per_step"] for r in rk["per_rank"]) - d["ms_per_step"]) < 1e-6
    assert all(k in rk["per_rank"][0] for k in ("device_index", "pci_bus_id", "device_name", "host", "local_rank"))


import pytest  # noqa: E402


@pytest.mark.parametrize("mode", ["sample", "train"])
def test_bench_bare_form_launches_its_own_ranks(mode):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts torch.distributed.run itself (as a
    child, before any GPU call), relays rank 0's single JSON line and exits with the workers' code.  `--mode train`: every
    rank reaches the step's collective."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OARD_BENCH_BACKEND="gloo", OARD_BENCH_DRY="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--mode", mode], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout       # stdout is exactly the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["dry_run"]
    assert d["metric"] == ("training_steps_per_sec" if mode == "train" else "denoising_steps_per_sec")
    assert d["ms_per_step"] >= 3.9


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, OARD_BENCH_DRY="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_bench_train_mode_two_ranks_under_the_launcher():
    env = dict(os.environ, OARD_BENCH_BACKEND="gloo", OARD_BENCH_DRY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--mode", "train"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["metric"] == "training_steps_per_sec" and "all-reduce" in d["config"]["parallelism"]
    rk = d["ranks"]                                          # train mode: the collective's own time and size, per rank
    assert rk["world_size"] == 2 and len(rk["per_rank"]) == 2
    assert all(r["all_reduce_bytes"] == 4096 and r["all_reduce_ms_mean"] > 0 for r in rk["per_rank"])


def test_bench_single_process_dry_run():
    env = dict(os.environ, OARD_BENCH_DRY="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["steps"] == 2


@pytest.mark.parametrize("mode", ["sample", "train"])
def test_bench_eight_ranks_gloo_dry_run(mode):
    """The driver's 8-GPU launch line (BASELINE configs[2] / [3]) on CPU: eight gloo ranks under torch.distributed.run.  What the record
    must show without the node: the world size the BACKEND reports, eight ranks with eight distinct LOCAL_RANK -> device bindings, eight
    distinct input seeds, eight pids, and `value` = world x B x steps / MAX over the ranks of the wall clock.  `train`: every one of
    the eight ranks reaches the step's single all-reduce (its result is checked inside the dry step) and reports its size and time."""
    env = dict(os.environ, OARD_BENCH_BACKEND="gloo", OARD_BENCH_DRY="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
           "--mode", mode]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dry_run"]
    rk = d["ranks"]
    assert rk["backend"] == "gloo" and rk["world_size"] == 8 and rk["expected_world_size"] == 8
    per = rk["per_rank"]
    assert [r["rank"] for r in per] == list(range(8))
    assert sorted(r["local_rank"] for r in per) == list(range(8))
    assert [r["bound_device_index"] for r in per] == [r["local_rank"] for r in per] and len({r["bound_device_index"] for r in per}) == 8
    assert len({r["input_seed"] for r in per}) == 8 and len({r["pid"] for r in per}) == 8
    # rank r sleeps 2 (r + 1) ms per step: the slowest is rank 7 with >= 16 ms, and the line's clock is the MAX over the ranks
    if mode == "sample":                                     # no collective inside the step: the ranks' own times differ
        assert rk["slowest_rank"] == 7 and per[7]["busy_ms_per_step"] >= 15.9 > per[0]["busy_ms_per_step"]
    else:                                                    # one all-reduce per step: every rank runs at the slowest rank's pace
        assert all(r["busy_ms_per_step"] >= 15.9 for r in per)
    assert d["ms_per_step"] >= 15.9 and abs(max(r["ms_per_step"] for r in per) - d["ms_per_step"]) < 1e-6
    assert abs(d["value"] - 8 * 64 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6      # whole-job aggregate over 8 ranks
    if mode == "train":
        assert d["metric"] == "training_steps_per_sec" and "dp8" in d["config"]["parallelism"]
        assert all(r["all_reduce_bytes"] == 4096 and r["all_reduce_ms_mean"] > 0 for r in per)
    else:
        assert d["metric"] == "denoising_steps_per_sec" and "replica x8" in d["config"]["parallelism"]
