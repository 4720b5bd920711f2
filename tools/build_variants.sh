#!/bin/bash
# builds experimental copies of the library with -D switches (on the GPU box) and A/Bs them
cd oareactdiff_amd/csrc
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result $flags oard_hip.hip -o /tmp/liboard_$i.so 2>&1 | grep -E "error" &
done
wait
cd ../..
i=0
for flags in "$@"; do
  i=$((i+1))
  echo "##### build $i: $flags"
  for v in ${VARIANTS:-3:2}; do
    g=${v%%:*}; q=${v##*:}
    OARD_LIB=/tmp/liboard_$i.so OARD_GCL_VARIANT=$g OARD_EQUI_VARIANT=$q python tools/gpu_debug.py g2s_prod_b1_n5 2>&1 | grep -E "l0.edgeweight|FINAL"
    OARD_LIB=/tmp/liboard_$i.so OARD_GCL_VARIANT=$g OARD_EQUI_VARIANT=$q python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   variant $v ms/step', round(d['ms_per_step'],2), r['families_ms_per_step'])"
  done
done
