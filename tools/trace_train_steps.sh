# launches and kernel time of ONE steady-state training step: tools/trace_train_steps.sh <tag>   (two kernel traces, 3 and 13 timed steps)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OARD_TRAIN_DUAL=0
for n in 3 13; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_tt$n -o t -- python bench.py --mode train --warmup 1 --steps $n > gpurun_out/${tag}_tt$n.log 2>&1
  python tools/prof_summary.py gpurun_out/${tag}_tt$n/t_results.db > gpurun_out/${tag}_tt$n.txt
  rm -rf gpurun_out/${tag}_tt$n
done
python tools/launch_diff.py gpurun_out/${tag}_tt3.txt gpurun_out/${tag}_tt13.txt 10 > gpurun_out/${tag}_train_launches_per_step.txt
head -70 gpurun_out/${tag}_train_launches_per_step.txt
