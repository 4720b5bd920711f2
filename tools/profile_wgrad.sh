#!/bin/bash
# kernel trace + SQ counter pass of tools/wgrad_probe.py (the five long weight-gradient products in isolation)
# usage: tools/profile_wgrad.sh <tag> [env assignments...]
tag=${1:-wg}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
p=gpurun_out/${tag}
timeout 300 rocprofv3 --kernel-trace --stats -d ${p}_trace -o t -- python tools/wgrad_probe.py 5 > ${p}_trace.log 2>&1
python tools/prof_summary.py ${p}_trace/t_results.db > ${p}_kernel_trace_summary.txt
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d ${p}_sq -o p -- python tools/wgrad_probe.py 5 > ${p}_sq.log 2>&1
python tools/pmc_summary.py ${p}_sq/p_results.db k_wgrad_t16 k_wgrad_lds "k_wgrad<" > ${p}_pmc_sq.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${p}_fetch -o p -- python tools/wgrad_probe.py 5 > ${p}_fetch.log 2>&1
python tools/pmc_summary.py ${p}_fetch/p_results.db k_wgrad_t16 k_wgrad_lds "k_wgrad<" > ${p}_pmc_fetch.txt
rm -rf ${p}_trace ${p}_sq ${p}_fetch
grep -E "k_wgrad|k_bgrad" ${p}_kernel_trace_summary.txt | cut -c1-150; cat ${p}_pmc_sq.txt ${p}_pmc_fetch.txt; cat ${p}_trace.log | tail -6
