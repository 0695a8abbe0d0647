#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -k "head or rollout or generate" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python tools/prof_rollout.py --cases 64 --hidden 32 --height 66 --width 65 2>&1 | grep "k_head_fwd\|of kernels"
python tools/prof_rollout.py --cases 256 --hidden 32 --height 66 --width 65 2>&1 | grep "k_head_fwd\|of kernels"
python tools/prof_rollout.py --cases 64 --hidden 20 --height 64 --width 64 2>&1 | grep "k_head_fwd\|of kernels"
python tools/kbench.py --only head_fwd --reps 30 2>&1 | grep head
python tools/kbench.py --only head_fwd --reps 30 --hidden 32 2>&1 | grep head
