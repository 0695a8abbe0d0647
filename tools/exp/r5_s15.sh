#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s15; mkdir -p $O
for hb in -1 512 384 268 256; do
  echo "== head_blocks=$hb"
  CFD_HEAD_BLOCKS=$hb python tools/prof_rollout.py --cases 64 --hidden 32 --height 66 --width 65 2>&1 | grep "k_head_fwd\|of kernels"
  CFD_HEAD_BLOCKS=$hb python tools/prof_rollout.py --cases 64 --hidden 20 --height 64 --width 64 2>&1 | grep "k_head_fwd\|of kernels"
done
