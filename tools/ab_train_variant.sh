#!/bin/bash
# A/B of debug options on the training step (single-stream sweep: additive kernel times): tools/ab_train_variant.sh "ENV=val" ...
for cfg in "$@"; do
  for i in 1 2; do
    env OARD_TRAIN_DUAL=0 $cfg python bench.py --mode train --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['train_step']
print('cfg [$cfg]', 'ms/step', round(d['ms_per_step'],2), t['families_ms_per_step'])"
  done
done
