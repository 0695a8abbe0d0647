#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s12; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "eight_waves or 25_to_32 or width_32" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python tools/exp/ab_step.py "head_waves=8" "" --rounds 3 --prof 2>&1 | grep -v amdgpu | tee $O/ab_c20.txt | head -8
python tools/exp/ab_step.py "head_waves=8" "" --rounds 3 --prof --hidden 32 2>&1 | grep -v amdgpu | tee $O/ab_c32.txt | head -8
python tools/exp/ab_step.py "head_waves=8,act_pieces=2" "act_pieces=2" --rounds 3 --prof 2>&1 | grep -v amdgpu | tee $O/ab_c20_ap2.txt | head -6
python tools/exp/ab_step.py "head_waves=8" "" --rounds 2 --hw 66 65 2>&1 | grep -v amdgpu | tee $O/ab_66.txt | head -3
