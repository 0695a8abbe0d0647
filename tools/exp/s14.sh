#!/bin/bash
# session 14: general fused FnoBlock (66x65): suite, A/B, per-kernel profile, rollout at C=20
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s14
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s14/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s14/pytest_gpu.log
python tools/exp/ab_step.py "block_gen=0" "" --hw 66 65 --rounds 3 --prof > gpurun_out/s14/ab_66x65.txt 2>&1; cat gpurun_out/s14/ab_66x65.txt
python tools/exp/ab_step.py "" --hw 66 65 --rounds 1 --prof 2>&1 | tail -22
python tools/exp/ab_step.py "" --rounds 2 2>&1 | tail -3
for g in 0 1; do CFD_BLOCK_GEN=$g python tools/prof_rollout.py --cases 64 --hidden 20 --height 66 --width 65 2>&1 | grep -v amdgpu | head -12; done
