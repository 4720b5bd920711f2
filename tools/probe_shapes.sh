cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 0 2; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/p2_$k -o t -- python tools/wgrad_probe.py 6 $k > gpurun_out/p2_$k.log 2>&1
python tools/prof_summary.py gpurun_out/p2_$k/t_results.db | grep -E "k_wg" | cut -c1-150
rm -rf gpurun_out/p2_$k
done
