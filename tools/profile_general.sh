#!/bin/bash
# rocprofv3 passes of the general-edge-list path (csrc/oard_general.h) on the headline batch's complete graph: kernel trace, then the
# matrix-pipe counters in a pass of their own.  usage (GPU box): tools/profile_general.sh <tag>    (summaries in gpurun_out/<tag>_general_*.txt)
tag=${1:-round6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OARD_GENERAL_TIME_SKIP_THREADS=1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_gtrace -o t -- python3 tools/general_time.py 64 > gpurun_out/${tag}_general_time.txt 2>&1
python tools/prof_summary.py gpurun_out/${tag}_gtrace/t_results.db > gpurun_out/${tag}_general_kernel_trace_summary.txt
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d gpurun_out/${tag}_gsq -o p -- python3 tools/general_time.py 64 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_gsq/p_results.db k_general_gemm_f64 > gpurun_out/${tag}_general_pmc_sq.txt
rm -rf gpurun_out/${tag}_gtrace gpurun_out/${tag}_gsq
tail -n 1 gpurun_out/${tag}_general_time.txt; head -n 14 gpurun_out/${tag}_general_kernel_trace_summary.txt; cat gpurun_out/${tag}_general_pmc_sq.txt
