#!/bin/bash
# Soak of the GPU suite (run on the GPU box): `pytest -m gpu` N times under each of the three environments the library ships for -
# the suite's pinned launch shapes (default), the library's own launch-shape heuristics (OARD_TEST_SHAPES=auto), the split-precision
# edge kernels (OARD_GCL_B3 / OARD_EQUI_B3 / OARD_TRAIN_B3) - plus the first-generation kernels on an experiment build.
# An ordering bug (a missing stream join, a scratch buffer reused too early) shows up as a run that fails once in a few: one line
# per run, pass / fail counts, appended to gpurun_out/<tag>_soak.txt.
# usage: tools/soak_tests.sh [tag] [runs per environment] [environments: default shapes_auto split_precision experiment | all]
# (gpurun limits one call to an hour: 3 environments x 3 runs of ~7 min do not fit in one call - pick the environments per call)
tag=${1:-round6}; n=${2:-5}; shift $(( $# < 2 ? $# : 2 )); which=" ${*:-all} "     # (`shift 2` with ONE argument shifts nothing: the tag became the selection)
want() { [[ "$which" == *" all "* || "$which" == *" $1 "* ]]; }
if ! [[ "$n" =~ ^[0-9]+$ ]] || [ "$n" -lt 1 ]; then echo "soak_tests.sh: runs per environment must be a positive number, got '$n'" >&2; exit 2; fi
sel=0; for e in default shapes_auto split_precision experiment; do want $e && sel=1; done
if [ $sel = 0 ]; then echo "soak_tests.sh: no environment selected by '$which' (default shapes_auto split_precision experiment | all)" >&2; exit 2; fi
out=gpurun_out/${tag}_soak.txt
mkdir -p gpurun_out
echo "# sources $(python -c 'import bench; print(bench.source_stamp())'), $(date -u +%Y-%m-%dT%H:%MZ)" >> $out
run() {   # name, env...
  local name=$1; shift
  for i in $(seq 1 $n); do
    env "$@" timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${tag}_soak_last.log 2>&1
    echo "$name run $i: $(tail -1 gpurun_out/${tag}_soak_last.log)" | tee -a $out
    grep -E "^(FAILED|ERROR) " gpurun_out/${tag}_soak_last.log | cut -c1-300 | sed "s/^/    /" | tee -a $out      # which tests, if any
  done
}
want default && run default OARD_SOAK=1
want shapes_auto && run shapes_auto OARD_TEST_SHAPES=auto
want split_precision && run split_precision OARD_GCL_B3=1 OARD_EQUI_B3=1 OARD_TRAIN_B3=1
want experiment || exit 0
# the first-generation (v0) kernels only exist in an experiment build: build it here (hipcc is on the box), run the variant test on it
exp=$GRAFT_REPO_ROOT/oareactdiff_amd/csrc/liboard_exp.so
if [ ! -f $exp ]; then
  OARD_LIB=$exp OARD_CXXFLAGS=-DOARD_EXPERIMENTS timeout 900 python -m oareactdiff_amd.build > gpurun_out/${tag}_exp_build.log 2>&1
fi
line=$(OARD_LIB=$exp timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "every_kernel_variant" -p no:cacheprovider 2>&1 | tail -1)
echo "experiment build, test_every_kernel_variant_is_parity_green: $line" | tee -a $out
