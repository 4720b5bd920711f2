#!/bin/bash
cd oareactdiff_amd/csrc
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result $flags oard_hip.hip -o /tmp/liboard_$i.so 2>&1 | grep -E "error" &
done
wait
cd ../..
i=0
export OARD_BENCH_ALLOW_NAN=1
for flags in "$@"; do
  i=$((i+1))
  echo "##### build $i: $flags"
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  OARD_LIB=/tmp/liboard_$i.so rocprofv3 --kernel-trace -d gpurun_out/abl_$i -o t -- python bench.py --batch ${ABL_BATCH:-64} --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python tools/prof_summary.py gpurun_out/abl_$i/t_results.db | grep -E "k_equi_node|k_gcl_node|k_node_pre"
  rm -rf gpurun_out/abl_$i
done
