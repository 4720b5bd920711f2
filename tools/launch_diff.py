"""Launches per steady-state step from two kernel-trace summaries (tools/prof_summary.py) of the same command with K1 and K1 + D
timed steps: (calls_long - calls_short) / D per kernel - the set-up launches (weight upload, topology, first-call allocations) cancel.
usage: python tools/launch_diff.py <summary_short> <summary_long> <D>"""
import re
import sys


def read(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$", line.rstrip())
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out


a, b, d = read(sys.argv[1]), read(sys.argv[2]), float(sys.argv[3])
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    if cb != ca:
        rows.append((k, (cb - ca) / d, (tb - ta) / d))
torchish = [r for r in rows if "at::native" in r[0] or "rocclr" in r[0] or "Cijk" in r[0]]
print(f"per steady-state step: {sum(r[1] for r in rows):.1f} launches, {sum(r[2] for r in rows):.2f} ms of kernel time; "
      f"torch / runtime kernels (at::native, rocclr copy / fill, hipBLASLt): {sum(r[1] for r in torchish):.1f} launches, "
      f"{sum(r[2] for r in torchish):.3f} ms; hipBLASLt GEMMs: {sum(r[1] for r in rows if 'Cijk' in r[0]):.0f}")
print(f"{'kernel':100s} {'launches/step':>14s} {'ms/step':>9s}")
for k, n, t in sorted(rows, key=lambda r: -r[2]):
    print(f"{k[:100]:100s} {n:14.1f} {t:9.3f}")
