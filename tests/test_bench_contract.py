"""bench.py prints ONE JSON line with the keys the driver reads (task statement, section 4)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], typ), k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["metric"] == "denoising_steps_per_sec" and d["unit"] == "reaction-steps/s" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
