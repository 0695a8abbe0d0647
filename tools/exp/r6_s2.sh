#!/bin/bash
# round 6, session 2: where the time of k_modes_mfma goes -- variants with parts of the kernel removed (modes.hip CFD_MM_EXP)
cd $GRAFT_REPO_ROOT
for v in base exp1 exp3 exp4 exp7 exp16 exp23; do
  if [ $v = base ]; then unset CFDBENCH_AMD_LIB; else export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_$v.so; fi
  echo "== $v"; python tools/kbench.py --only mix,mix_adj_wgrad,adam,loss_scores --reps 50 2>/dev/null | grep -v amdgpu.ids | tail -4
done
