cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r3_train_b.json 2> gpurun_out/r3_train_b.err
python bench.py --mode train --batch 14 --steps 10 --warmup 3 > gpurun_out/r3_train_b14.json 2>> gpurun_out/r3_train_b.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r3t_ttrace -o t -- python bench.py --mode train --steps 3 --warmup 1 > gpurun_out/r3t_ttrace.log 2>&1
python tools/prof_summary.py gpurun_out/r3t_ttrace/t_results.db > gpurun_out/r3t_train_kernel_trace_summary.txt
rm -rf gpurun_out/r3t_ttrace
OARD_TRAIN_PROFILE=1 python bench.py --mode train --steps 3 --warmup 1 2>&1 | grep "backward sweep" | tail -3
