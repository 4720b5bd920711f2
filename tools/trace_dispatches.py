"""Per-dispatch durations of the kernels matching a substring, in launch order, from a rocprofv3 --kernel-trace rocpd database:
python tools/trace_dispatches.py <db> <substring>      (is a trace average pulled up by a few cold launches, or uniformly higher?)"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else "kernel_name"
rows = [(n, (b - a) / 1e3) for n, a, b in c.execute(f"select s.{namecol}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start") if sys.argv[2] in n]
print(len(rows), "dispatches;", " ".join(f"{us:.0f}" for _, us in rows))
