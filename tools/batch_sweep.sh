#!/bin/bash
# denoising step time over batch sizes (run on the GPU box): tools/batch_sweep.sh [B ...]
for b in "${@:-1 2 4 8 16 32 64 128 256 512}"; do
  for bb in $b; do
    python bench.py --batch $bb --steps 20 --warmup 3 --no-cpu-baseline --quick $OARD_SWEEP_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B', $bb, 'ms/step', round(d['ms_per_step'],3), 'reaction-steps/s', round(d['value'],1))"
  done
done
