#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s5; mkdir -p $O
python tools/prof_rollout.py --cases 64 --hidden 32 --height 66 --width 65 2>&1 | grep -v amdgpu | tee $O/rollout_c32_66x65_64.txt
python tools/prof_rollout.py --cases 256 --hidden 32 --height 66 --width 65 2>&1 | grep -v amdgpu | tee $O/rollout_c32_66x65_256.txt
python tools/prof_rollout.py --cases 64 --hidden 32 --height 66 --width 65 --dtype bf16 2>&1 | grep -v amdgpu | tee $O/rollout_c32_66x65_64_bf16.txt
python tools/exp/ab_step.py "" --rounds 2 --prof --batch 8 2>&1 | grep -v amdgpu | tee $O/ab_b8.txt
python tools/exp/ab_step.py "" --rounds 2 --prof --batch 256 --hidden 32 2>&1 | grep -v amdgpu | tee $O/ab_c32.txt
