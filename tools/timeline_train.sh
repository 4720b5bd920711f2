# concurrency picture of the training step under the default two-stream schedule: tools/timeline_train.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/${tag}_tl -o t -- python bench.py --mode train --warmup 2 --steps 6 > gpurun_out/${tag}_tl.log 2>&1
python tools/timeline.py gpurun_out/${tag}_tl/t_results.db > gpurun_out/${tag}_train_timeline.txt
rm -rf gpurun_out/${tag}_tl
cat gpurun_out/${tag}_train_timeline.txt
