#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s9; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-300
python tools/exp/ab_step.py "" --rounds 3 --prof 2>&1 | grep -v amdgpu | tee $O/ab.txt | head -22
python tools/exp/ab_step.py "" --rounds 2 --batch 8 2>&1 | grep -v amdgpu | tee $O/ab_b8.txt | head -3
