#!/bin/bash
# A/B of environment settings on the denoising step (run on the GPU box): tools/ab_env.sh "ENV=val ENV2=val" "..." ...
# prints ms/step of the default schedule and the isolated per-family kernel times of bench.py --quick, two runs per configuration
export OARD_BENCH_ALLOW_NAN=1
for cfg in "$@"; do
  for i in 1 2; do
    env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --quick 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$cfg]', 'ms/step', round(d['ms_per_step'],3), r['families_ms_per_step'], 'frac', round(r['frac'],3))"
  done
done
