# kernel trace of the training bench: tools/trace_train.sh <tag> [env assignments...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_ttrace -o t -- python bench.py --mode train --steps 3 --warmup 1 > gpurun_out/${tag}_ttrace.log 2>&1
python tools/prof_summary.py gpurun_out/${tag}_ttrace/t_results.db > gpurun_out/${tag}_train_kernel_trace_summary.txt
rm -rf gpurun_out/${tag}_ttrace
grep -E "k_wgrad" gpurun_out/${tag}_train_kernel_trace_summary.txt
