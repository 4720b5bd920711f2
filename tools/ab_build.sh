#!/bin/bash
# build experimental copies of the library with -D switches (on the GPU box) and A/B the whole step:
#   ./tools/ab_build.sh "<batch sizes>" "<flags 1>" "<flags 2>" ...
bs=$1; shift
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
  OARD_LIB=/tmp/liboard_$i.so python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "forward_matches" 2>&1 | tail -1
  for b in $bs; do
    for r in 1 2; do
    OARD_LIB=/tmp/liboard_$i.so python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   B=$b ms/step', round(d['ms_per_step'],3), r['families_ms_per_step'])"
    done
  done
done
