#!/bin/bash
# training half of tools/profile.sh: usage tools/profile_train.sh <tag>
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# training step (BASELINE config 4): kernel traces of bench.py --mode train with 3 and with 13 timed steps (their difference / 10 = the launches
# of ONE steady-state step, free of the set-up launches: weight upload, topology, allocator warm-up), then separate PMC passes
unset OARD_PARTS
# OARD_TRAIN_DUAL=0: the sweep on ONE stream, so that kernel durations do not overlap and add up to the step (the default - weight
# gradients on a second stream - is what bench.py times: the bench line under the profiler is taken with the default)
T="python bench.py --mode train --warmup 1"
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/${tag}_train_bench_line.json 2>/dev/null
export OARD_TRAIN_DUAL=0
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_ttrace -o t -- $T --steps 3 > gpurun_out/${tag}_ttrace.log 2>&1
python tools/prof_summary.py gpurun_out/${tag}_ttrace/t_results.db > gpurun_out/${tag}_train_kernel_trace_summary.txt
grep '"metric"' gpurun_out/${tag}_ttrace.log | tail -1 > gpurun_out/${tag}_train_bench_line_under_profiler.json
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_ttrace13 -o t -- $T --steps 13 > gpurun_out/${tag}_ttrace13.log 2>&1
python tools/prof_summary.py gpurun_out/${tag}_ttrace13/t_results.db > gpurun_out/${tag}_train_kernel_trace_summary_13steps.txt
python tools/launch_diff.py gpurun_out/${tag}_train_kernel_trace_summary.txt gpurun_out/${tag}_train_kernel_trace_summary_13steps.txt 10 > gpurun_out/${tag}_train_launches_per_step.txt
rm -rf gpurun_out/${tag}_ttrace gpurun_out/${tag}_ttrace13
TK="k_wgrad k_gcl_edge_bwd k_equi_edge_bwd k_gcl_edge_p k_gcl_edge_v1 k_equi_edge_v1"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT -d gpurun_out/${tag}_tsq -o p -- $T --steps 3 > gpurun_out/${tag}_tsq.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_tsq/p_results.db --per-forward 1 $TK > gpurun_out/${tag}_train_pmc_sq.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_tfetch -o p -- $T --steps 3 > gpurun_out/${tag}_tfetch.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_tfetch/p_results.db --per-forward 1 $TK > gpurun_out/${tag}_train_pmc_fetch.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_twrite -o p -- $T --steps 3 > gpurun_out/${tag}_twrite.log 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_twrite/p_results.db --per-forward 1 $TK > gpurun_out/${tag}_train_pmc_write.txt
rm -rf gpurun_out/${tag}_tsq gpurun_out/${tag}_tfetch gpurun_out/${tag}_twrite
cat gpurun_out/${tag}_train_launches_per_step.txt | head -30; cat gpurun_out/${tag}_train_pmc_sq.txt gpurun_out/${tag}_train_pmc_fetch.txt gpurun_out/${tag}_train_pmc_write.txt
