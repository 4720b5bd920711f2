"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db or *_kernel_trace.csv): per-kernel
launch count, average and total duration.  Usage: python tools/prof_summary.py <file> [skip_first_n_per_kernel]"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("oard_general::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def rows_from_db(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in scols else "kernel_name"
    q = f"select s.{namecol}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    for name, a, b in c.execute(q):
        yield name, (b - a) / 1e3


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(list)
    for name, us in rows:
        agg[short(name)].append(us)
    tot = sum(sum(v[skip:]) for v in agg.values())
    print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][skip:])):
        v = v[skip:]
        if not v:
            continue
        print(f"{name:92s} {len(v):6d} {sum(v) / len(v):10.1f} {sum(v) / 1e3:10.3f} {100 * sum(v) / tot:6.2f}")


if __name__ == "__main__":
    main()
