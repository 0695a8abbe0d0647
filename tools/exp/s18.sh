#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s18
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s18/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s18/pytest_gpu.log
python tools/exp/ab_step.py "" --hw 66 65 --rounds 3 --prof 2>&1 | tail -20
python tools/prof_rollout.py --cases 64 2>&1 | grep -v amdgpu
