#!/bin/bash
# rocprofv3 passes of BASELINE configs[4] (B = 4 reactions x 3 x 128 atoms, fp32), both position scales, one sub-batch as tools/profile.sh;
# every profiler run under `timeout`.  bench.py's "config5" object reads the FETCH / WRITE summaries (HBM GB/s while the kernels run).
# usage: tools/profile_cfg5.sh <tag>     (run on the GPU box; summaries land in gpurun_out/<tag>_cfg5{n,x3}_*.txt)
tag=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OARD_PARTS=1
# only fixed-distribution steps under the profiler: the (quick) sampling leg runs a diverging trajectory whose inner edges leave the cutoff -
# EquiMessage then skips them, and its calls would pull the per-kernel averages of the trace down (round 5)
export OARD_BENCH_SKIP=sampler
for v in n x3; do
  scale=1; [ $v = x3 ] && scale=3
  B="python bench.py --batch 4 --atoms 128 --pos-scale $scale --steps 4 --warmup 2 --no-cpu-baseline --quick"
  p=gpurun_out/${tag}_cfg5${v}
  timeout 300 rocprofv3 --kernel-trace --stats -d ${p}_trace -o t -- $B > ${p}_trace.log 2>&1
  python tools/prof_summary.py ${p}_trace/t_results.db > ${p}_kernel_trace_summary.txt
  grep '"metric"' ${p}_trace.log | tail -1 > ${p}_bench_line_under_profiler.json
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d ${p}_sq -o p -- $B > ${p}_sq.log 2>&1
  python tools/pmc_summary.py ${p}_sq/p_results.db --per-forward 1 k_gcl_edge k_equi_edge k_equi_node k_gcl_node > ${p}_pmc_sq.txt
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${p}_fetch -o p -- $B > ${p}_fetch.log 2>&1
  python tools/pmc_summary.py ${p}_fetch/p_results.db --per-forward 1 k_gcl_edge k_equi_edge k_equi_node k_gcl_node > ${p}_pmc_fetch.txt
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${p}_write -o p -- $B > ${p}_write.log 2>&1
  python tools/pmc_summary.py ${p}_write/p_results.db --per-forward 1 k_gcl_edge k_equi_edge k_equi_node k_gcl_node > ${p}_pmc_write.txt
  rm -rf ${p}_trace ${p}_sq ${p}_fetch ${p}_write
  head -10 ${p}_kernel_trace_summary.txt | cut -c1-140; cat ${p}_pmc_sq.txt ${p}_pmc_fetch.txt ${p}_pmc_write.txt
done
