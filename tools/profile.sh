#!/bin/bash
# rocprofv3 passes for one round: kernel trace + separate PMC passes (never combined with sys/hip traces).
# The batch is run as ONE sub-batch (OARD_PARTS=1): whole-batch launches, nothing overlapping — the same
# setting bench.py's roofline leg uses to time the kernels.  (The default schedule splits B=64 into 4
# concurrent sub-batches, which is faster end to end but makes per-kernel durations overlap.)
# usage: tools/profile.sh <tag>     (run on the GPU box; summaries land in gpurun_out/<tag>_*.txt)
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OARD_PARTS=1
# only fixed-distribution steps under the profiler: the (quick) sampling leg runs a diverging trajectory whose inner edges leave the cutoff -
# EquiMessage then skips them, and its calls would pull the per-kernel averages of the trace down (round 5)
export OARD_BENCH_SKIP=sampler
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --quick"
python -c "import bench; print(bench.source_stamp())" > gpurun_out/${tag}_source_stamp.txt
P=1    # sub-batches in this profiling configuration
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o t -- $B > gpurun_out/${tag}_trace.log 2>&1
python tools/prof_summary.py gpurun_out/${tag}_trace/t_results.db > gpurun_out/${tag}_kernel_trace_summary.txt
grep '"metric"' gpurun_out/${tag}_trace.log | tail -1 > gpurun_out/${tag}_bench_line_under_profiler.json
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT -d gpurun_out/${tag}_sq -o p -- $B > gpurun_out/${tag}_sq.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_sq/p_results.db --per-forward $P k_gcl_edge k_equi_edge k_equi_node k_gcl_node > gpurun_out/${tag}_pmc_sq.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch -o p -- $B > gpurun_out/${tag}_fetch.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_fetch/p_results.db --per-forward $P k_gcl_edge k_equi_edge > gpurun_out/${tag}_pmc_fetch.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_write -o p -- $B > gpurun_out/${tag}_write.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_write/p_results.db --per-forward $P k_gcl_edge k_equi_edge > gpurun_out/${tag}_pmc_write.txt
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_sq gpurun_out/${tag}_fetch gpurun_out/${tag}_write
head -14 gpurun_out/${tag}_kernel_trace_summary.txt; cat gpurun_out/${tag}_pmc_sq.txt gpurun_out/${tag}_pmc_fetch.txt gpurun_out/${tag}_pmc_write.txt
bash tools/profile_train.sh $tag
