"""Per-kernel PMC counter sums from a rocprofv3 --pmc run (rocpd sqlite db).
Usage: python tools/pmc_summary.py <db> [--per-forward PARTS] [kernel-substring ...]
With --per-forward the sums are divided by the number of forward calls in the run
(= launches of k_post / PARTS), giving per-denoising-step totals per kernel family."""
import re
import sqlite3
import sys
from collections import defaultdict

args = sys.argv[1:]
path = args.pop(0)
parts = None
if args and args[0] == "--per-forward":
    parts = int(args[1]); args = args[2:]
subs = args or ["k_gcl_edge", "k_equi_edge"]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else "kernel_name"
clean = lambda n: re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "").replace("oard_general::", "")).replace("void ", "")
n_post = sum(1 for (n,) in c.execute(f"select s.{namecol} from {kd} d join {ks} s on d.kernel_id = s.id") if clean(n).strip() == "k_post")
forwards = n_post / parts if parts else None
q = (f"select s.{namecol}, d.start, d.end, p.name, e.value, d.id from {pe} e join {kd} d on e.event_id = d.event_id "
     f"join {ks} s on d.kernel_id = s.id join {pi} p on e.pmc_id = p.id")
agg = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
dur = defaultdict(dict)
for name, a, b, cname, val, did in c.execute(q):
    name = clean(name)
    fam = next((s for s in subs if s in name), None)
    if fam is None:
        continue
    key = fam if parts else name
    agg[key][cname][did] += val
    dur[key][did] = (b - a) / 1e3
if parts:
    print(f"# forward calls in this run: {forwards:.0f} (k_post launches {n_post} / {parts} sub-batches); values are per forward call (one denoising step)")
for name, cs in sorted(agg.items()):
    d = list(dur[name].values())
    if parts:
        print(f"== {name}  launches/forward={len(d) / forwards:.1f}  kernel_ms/forward={sum(d) / 1e3 / forwards:.3f}")
    else:
        print(f"== {name}  launches={len(d)} avg_us={sum(d) / len(d):.1f}")
    for cname, per in sorted(cs.items()):
        v = list(per.values())
        if parts:
            print(f"   {cname:34s} per forward = {sum(v) / forwards:.5g}")
        else:
            print(f"   {cname:34s} avg/launch = {sum(v) / len(v):.4g}")
