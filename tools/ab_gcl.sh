#!/bin/bash
# A/B of GCL edge-kernel shapes in an experiment build (-DOARD_EXPERIMENTS; built here into csrc/liboard_exp.so):
#   tools/ab_gcl.sh 2 8 9 10 ...     -> whole-step ms and per-family kernel ms for each gcl_variant (run on the GPU box)
export OARD_LIB=$GRAFT_REPO_ROOT/oareactdiff_amd/csrc/liboard_exp.so
for v in "$@"; do
  for i in 1 2; do
    OARD_GCL_VARIANT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --quick 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('gcl_variant $v', 'ms/step', round(d['ms_per_step'],3), r['families_ms_per_step'], 'frac', round(r['frac'],3))"
  done
done
