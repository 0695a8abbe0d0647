#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s15
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s15/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s15/pytest_gpu.log
for c in 16 64; do for g in 0 1; do echo "block_gen=$g"; CFD_BLOCK_GEN=$g python tools/prof_rollout.py --cases $c --hidden 20 --height 66 --width 65 2>&1 | grep -v amdgpu | grep "cases\|k_block\|k_idft_add\|k_chanmix\|graph"; done; done
python tools/exp/ab_step.py "block_gen=0" "" --hw 66 65 --batch 8 --rounds 3 --steps 100 2>&1 | tail -3
python tools/exp/ab_step.py "block_gen=0" "" --hw 66 65 --batch 64 --rounds 3 --steps 50 2>&1 | tail -3
