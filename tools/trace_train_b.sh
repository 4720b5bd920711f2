#!/bin/bash
# kernel trace of the training step at batch size B, single-stream sweep, + step time with one / two streams: tools/trace_train_b.sh <B> <tag>
B=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dual in 1 0; do
  OARD_TRAIN_DUAL=$dual python bench.py --mode train --batch $B --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['train_step']
print('B $B dual $dual ms/step', round(d['ms_per_step'],2), t['families_ms_per_step'], 'kernels', t['hip_kernel_ms_per_step'])"
done
export OARD_TRAIN_DUAL=0
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_t3 -o t -- python bench.py --mode train --batch $B --warmup 1 --steps 3 > gpurun_out/${tag}_t3.log 2>&1
python tools/prof_summary.py gpurun_out/${tag}_t3/t_results.db > gpurun_out/${tag}_s3.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_t13 -o t -- python bench.py --mode train --batch $B --warmup 1 --steps 13 > gpurun_out/${tag}_t13.log 2>&1
python tools/prof_summary.py gpurun_out/${tag}_t13/t_results.db > gpurun_out/${tag}_s13.txt
python tools/launch_diff.py gpurun_out/${tag}_s3.txt gpurun_out/${tag}_s13.txt 10 > gpurun_out/${tag}_launches_per_step.txt
rm -rf gpurun_out/${tag}_t3 gpurun_out/${tag}_t13
head -45 gpurun_out/${tag}_launches_per_step.txt
