#!/bin/bash
# rocprofv3 passes of the split-precision second line (bench.py --precision bf16x3), one sub-batch as tools/profile.sh.
# (every profiler run under `timeout`: an aborted counter pass once sat in its finaliser until the box's limit)
# usage: tools/profile_b3.sh <tag>     (run on the GPU box; summaries land in gpurun_out/<tag>_bf16x3_*.txt)
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OARD_PARTS=1
# only fixed-distribution steps under the profiler: the (quick) sampling leg runs a diverging trajectory whose inner edges leave the cutoff -
# EquiMessage then skips them, and its calls would pull the per-kernel averages of the trace down (round 5)
export OARD_BENCH_SKIP=sampler
B="python bench.py --precision bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --quick"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_b3trace -o t -- $B > gpurun_out/${tag}_b3trace.log 2>&1
python tools/prof_summary.py gpurun_out/${tag}_b3trace/t_results.db > gpurun_out/${tag}_bf16x3_kernel_trace_summary.txt
grep '"metric"' gpurun_out/${tag}_b3trace.log | tail -1 > gpurun_out/${tag}_bf16x3_bench_line_under_profiler.json
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/${tag}_b3sq -o p -- $B > gpurun_out/${tag}_b3sq.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_b3sq/p_results.db --per-forward 1 k_gcl_edge k_equi_edge > gpurun_out/${tag}_bf16x3_pmc_sq.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_b3fetch -o p -- $B > gpurun_out/${tag}_b3fetch.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_b3fetch/p_results.db --per-forward 1 k_gcl_edge k_equi_edge > gpurun_out/${tag}_bf16x3_pmc_fetch.txt
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_b3write -o p -- $B > gpurun_out/${tag}_b3write.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_b3write/p_results.db --per-forward 1 k_gcl_edge k_equi_edge > gpurun_out/${tag}_bf16x3_pmc_write.txt
rm -rf gpurun_out/${tag}_b3trace gpurun_out/${tag}_b3sq gpurun_out/${tag}_b3fetch gpurun_out/${tag}_b3write
head -12 gpurun_out/${tag}_bf16x3_kernel_trace_summary.txt | cut -c1-140; cat gpurun_out/${tag}_bf16x3_pmc_sq.txt gpurun_out/${tag}_bf16x3_pmc_fetch.txt gpurun_out/${tag}_bf16x3_pmc_write.txt
