#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s21; mkdir -p $O
python tools/exp/ab_step.py "" "stem_fuse=0" --rounds 3 --prof 2>&1 | grep -v amdgpu | tee $O/ab.txt | grep "best\|k_block_bwd_stem\|k_stem_grad"
