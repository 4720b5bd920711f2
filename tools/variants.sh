#!/bin/bash
# A/B the edge-kernel variants: parity tests + per-family timing for each (run on the GPU box)
mkdir -p gpurun_out
for v in "$@"; do
  g=${v%%:*}; q=${v##*:}
  echo "=== gcl_variant=$g equi_variant=$q"
  OARD_GCL_VARIANT=$g OARD_EQUI_VARIANT=$q python -m pytest tests -m gpu -x -q 2>&1 | tail -3
  OARD_GCL_VARIANT=$g OARD_EQUI_VARIANT=$q python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step', round(d['ms_per_step'],2), 'families', r['families_ms_per_step'], 'dom', r['kernel'], round(r['achieved'],1), 'TF; other', r['other_kernel'])"
done
