#!/bin/bash
# timing of tools/wgrad_probe.py under ablation builds of the library: tools/ab_wgt.sh <suffix> ...  (csrc/liboard_<suffix>.so; "hip" = the product)
for s in "$@"; do
  export OARD_LIB=$GRAFT_REPO_ROOT/oareactdiff_amd/csrc/liboard_$s.so
  echo "== lib $s"; python tools/wgrad_probe.py 8 2>/dev/null | grep rows
done
