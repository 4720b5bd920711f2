# concurrency picture of the B = 64 denoising step under the default sub-batch schedule: tools/timeline_fwd.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/${tag}_tlf -o t -- python bench.py --warmup 3 --steps 12 --no-cpu-baseline --quick > gpurun_out/${tag}_tlf.log 2>&1
python tools/timeline.py gpurun_out/${tag}_tlf/t_results.db > gpurun_out/${tag}_fwd_timeline.txt
rm -rf gpurun_out/${tag}_tlf
cat gpurun_out/${tag}_fwd_timeline.txt
