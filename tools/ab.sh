#!/bin/bash
# A/B whole-step time: ./tools/ab.sh "ENV=.. ENV=.." "ENV=.." ...  (each argument = one environment, 3 bench runs each)
for e in "$@"; do
  for i in 1 2 3; do
    env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e', 'ms/step', round(d['ms_per_step'],3), r['families_ms_per_step'])"
  done
done
