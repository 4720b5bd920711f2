"""Concurrency picture of the default (sub-batch-parallel) schedule from a rocprofv3 --kernel-trace rocpd
database: for the steady-state forwards, how long k kernels were in flight at once, and per kernel family
the in-flight time.  Usage: python tools/timeline.py <results.db> [n_forwards_to_skip]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    c = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in scols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"s.{namecol}, d.start, d.end" + (f", d.{qcol}" if qcol else ", 0")
    rows = list(c.execute(f"select {sel} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    rows = [(re.sub(r"^void ", "", re.sub(r"[<(].*$", "", n)), a, b, q) for n, a, b, q in rows]
    # forwards are delimited by k_prep launches (first kernel of a part); group by gaps between k_post and next k_prep
    preps = [i for i, r in enumerate(rows) if r[0].startswith("k_prep")]
    if not preps:
        print("no k_prep found; columns:", cols)
        return
    t_first = rows[preps[len(preps) // 2]][1]     # second half = steady state
    rows = [r for r in rows if r[1] >= t_first]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    ev = []
    for n, a, b, q in rows:
        ev.append((a, 1, n))
        ev.append((b, -1, n))
    ev.sort()
    hist = defaultdict(float)
    fam_alone = defaultdict(float)
    active = defaultdict(int)
    k = 0
    last = t0
    for t, d, n in ev:
        hist[k] += t - last
        if k == 1:
            for name, cnt in active.items():
                if cnt:
                    fam_alone[name] += t - last
        last = t
        k += d
        active[n] += d
    wall = (t1 - t0) / 1e6
    print(f"window {wall:.3f} ms, {len(rows)} dispatches, queues: {sorted(set(r[3] for r in rows))}")
    for kk in sorted(hist):
        print(f"  {kk} kernels in flight: {hist[kk] / 1e6:8.3f} ms ({100 * hist[kk] / (t1 - t0):5.1f} %)")
    print("time with exactly ONE kernel in flight, by kernel:")
    for n, v in sorted(fam_alone.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {n:40s} {v / 1e6:8.3f} ms")
    # idle gaps (nothing in flight): total by the kernel that ends the gap, and the longest ones
    gaps = defaultdict(lambda: [0, 0.0])
    longest = []
    k = 0
    gap_start = None
    for t, d, n in ev:
        if d == 1 and k == 0 and gap_start is not None:
            g_ = t - gap_start
            gaps[n][0] += 1
            gaps[n][1] += g_
            longest.append((g_, n))
        k += d
        if k == 0:
            gap_start = t
    print("idle gaps by the kernel that ends them: count, total ms, avg us")
    for n, (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {n:40s} {cnt:6d} {tot / 1e6:8.3f} {tot / cnt / 1e3:8.1f}")
    print("longest gaps (us):", ", ".join(f"{g_ / 1e3:.0f} before {n}" for g_, n in sorted(longest, reverse=True)[:10]))
    agg = defaultdict(list)
    for n, a, b, q in rows:
        agg[n].append((b - a) / 1e3)
    print("per kernel (under concurrency): calls, avg us, total ms")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"  {n:40s} {len(v):6d} {sum(v) / len(v):10.1f} {sum(v) / 1e3:10.3f}")


if __name__ == "__main__":
    main()
