"""Per-kernel PMC counter averages from a rocprofv3 --pmc run (rocpd sqlite db).
Usage: python tools/pmc_summary.py <db> [kernel-substring ...]"""
import re
import sqlite3
import sys
from collections import defaultdict

path = sys.argv[1]
subs = sys.argv[2:] or ["k_gcl_edge", "k_equi_edge"]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else "kernel_name"
pcols = [r[1] for r in c.execute(f"pragma table_info({pe})")]
q = (f"select s.{namecol}, d.start, d.end, p.name, e.value, d.id from {pe} e join {kd} d on e.event_id = d.event_id "
     f"join {ks} s on d.kernel_id = s.id join {pi} p on e.pmc_id = p.id")
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for name, a, b, cname, val, did in c.execute(q):
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    if not any(s in name for s in subs):
        continue
    agg[name][cname].append((did, val))
    dur[name][did] = (b - a) / 1e3
for name, cs in agg.items():
    d = list(dur[name].values())
    print(f"== {name}  launches={len(d)} avg_us={sum(d) / len(d):.1f}")
    for cname, vals in sorted(cs.items()):
        per = defaultdict(float)
        for did, v in vals:
            per[did] += v
        v = list(per.values())
        print(f"   {cname:34s} avg/launch = {sum(v) / len(v):.4g}")
