#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s14; mkdir -p $O
for w in 256 128 384 512 768 256; do
  CFD_WGRAD_WG=$w python tools/exp/ab_step.py "" --rounds 2 --prof 2>&1 | grep -v amdgpu > $O/ab_wg$w.txt
  echo "wgrad_wg=$w: $(head -1 $O/ab_wg$w.txt | cut -c40-90) $(grep mixadj $O/ab_wg$w.txt)"
done
