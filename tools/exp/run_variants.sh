#!/bin/bash
# GPU box: time kernel variants built into tools/exp/libs (dev experiment).  usage: run_variants.sh "<kbench --only list>" <exp ids...>
cd $GRAFT_REPO_ROOT
ONLY=$1; shift
for e in "$@"; do
  cp tools/exp/libs/lib_exp$e.so cfdbench_amd/_C/libcfdbench_amd.so
  echo "== CFD_EXP=$e"; timeout 120 python tools/kbench.py --only $ONLY 2>&1 | grep -v amdgpu.ids
done
cp tools/exp/libs/lib_exp0.so cfdbench_amd/_C/libcfdbench_amd.so
