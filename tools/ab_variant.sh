#!/bin/bash
# A/B of kernel variants selected by debug options (run on the GPU box): tools/ab_variant.sh "ENV=val ..." "ENV=val ..."
for cfg in "$@"; do
  for i in 1 2; do
    env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --quick 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg [$cfg]', 'ms/step', round(d['ms_per_step'],3), r['families_ms_per_step'], 'frac', round(r['frac'],3))"
  done
done
