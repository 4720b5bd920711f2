#!/bin/bash
# A/B of whole-library builds on the training step: tools/ab_train_lib.sh <suffix> ...  ->  csrc/liboard_<suffix>.so  (OARD_TRAIN_DUAL=0: additive kernel times)
for s in "$@"; do
  export OARD_LIB=$GRAFT_REPO_ROOT/oareactdiff_amd/csrc/liboard_$s.so
  for i in 1 2; do
    OARD_TRAIN_DUAL=0 python bench.py --mode train --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['train_step']
print('lib $s', 'ms/step', round(d['ms_per_step'],2), t['families_ms_per_step'], 'loss', round(t['loss'],4))"
  done
done
