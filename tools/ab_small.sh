#!/bin/bash
# small-batch A/B: ./tools/ab_small.sh "B1 B2 .." "ENV=.. ENV=.." ...
bs=$1; shift
for e in "$@"; do
  for b in $bs; do
    env $e python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e', 'B=$b', 'ms/step', round(d['ms_per_step'],3), r['families_ms_per_step'])"
  done
done
