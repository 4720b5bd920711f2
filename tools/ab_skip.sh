#!/bin/bash
# Marginal cost of each kernel family INSIDE the concurrent step (run on the GPU box): the step is timed with the launches of one
# family dropped (debug option skip_families, results are garbage).  tools/ab_skip.sh [mask ...]   (bit: 0 gcl_edge, 1 equi_edge, 2 node, 3 init)
export OARD_BENCH_ALLOW_NAN=1
for m in "${@:-0 1 2 4 8 12}"; do
  for mm in $m; do
  export OARD_SKIP_FAMILIES=$mm
  for i in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --quick 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('skip mask $mm', 'ms/step', round(d['ms_per_step'],3))"
  done
  done
done
