#!/bin/bash
# A/B of experiment / ablation builds of the whole library (run on the GPU box): tools/ab_lib.sh <suffix> ...  ->  csrc/liboard_<suffix>.so
export OARD_BENCH_ALLOW_NAN=1
for s in "$@"; do
  export OARD_LIB=$GRAFT_REPO_ROOT/oareactdiff_amd/csrc/liboard_$s.so
  for i in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --quick 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lib $s', 'ms/step', round(d['ms_per_step'],3), r['families_ms_per_step'], 'frac', round(r['frac'],3))"
  done
done
