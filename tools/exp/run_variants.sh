#!/bin/bash
# GPU box: time the k_block variants built into tools/exp/libs (dev experiment)
cd $GRAFT_REPO_ROOT
for e in 0 1 2 3 7; do
  cp tools/exp/libs/lib_exp$e.so cfdbench_amd/_C/libcfdbench_amd.so
  echo "== CFD_EXP=$e"; timeout 120 python tools/kbench.py --only block_fwd,block_fwd_act,block_bwd,block_bwd_dgelu 2>&1 | grep block
done
cp tools/exp/libs/lib_exp0.so cfdbench_amd/_C/libcfdbench_amd.so
