#!/bin/bash
# Dev experiment helper (GPU box): time kernel variants that were built with different -D switches.
#   build (CPU container):  for e in 1 2; do python -c "from cfdbench_amd.build import build; build(force=True, extra_flags=['-DCFD_EXP=$e'])"; \
#                             mkdir -p tools/exp/libs; cp cfdbench_amd/_C/libcfdbench_amd.so tools/exp/libs/lib_exp$e.so; done
#                           (and an unmodified build as lib_exp0.so; tools/exp/libs/ is git-ignored via *.so)
#   run:    gpurun -- 'bash tools/exp/run_variants.sh "<kbench --only list>" 0 1 2'
cd $GRAFT_REPO_ROOT
ONLY=$1; shift
for e in "$@"; do
  cp tools/exp/libs/lib_exp$e.so cfdbench_amd/_C/libcfdbench_amd.so
  echo "== variant $e"; timeout 120 python tools/kbench.py --only $ONLY 2>&1 | grep -v amdgpu.ids
done
cp tools/exp/libs/lib_exp0.so cfdbench_amd/_C/libcfdbench_amd.so
