#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -k "head or engine or fno or three_piece or two_piece" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/exp/ab_step.py "" --rounds 3 --prof 2>&1 | grep -v amdgpu | tee $O/ab.txt | head -12
bash tools/pmc_traffic.sh r5s8 head python $GRAFT_REPO_ROOT/tools/kbench.py --only head_train --reps 10 > $O/head_traffic.log 2>&1; grep k_head $O/head_traffic.log
