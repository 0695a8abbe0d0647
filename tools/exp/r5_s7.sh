#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "25_to_32 or width_32 or block" > $O/pytest_block.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_block.log
python tools/kbench.py --hidden 32 --only block,chanmix,idft_add 2>&1 | grep -v amdgpu | tee $O/kbench_c32_64.txt
python tools/kbench.py --hidden 32 --height 66 --width 65 --only block,chanmix,idft_add 2>&1 | grep -v amdgpu | tee $O/kbench_c32_66.txt
python tools/kbench.py --hidden 32 --height 66 --width 65 --batch 64 --only block,chanmix,idft_add 2>&1 | grep -v amdgpu | tee $O/kbench_c32_66_b64.txt
python tools/prof_rollout.py --cases 64 --hidden 32 --height 66 --width 65 2>&1 | grep -v amdgpu | tee $O/rollout_c32_66x65_64.txt
python tools/exp/ab_step.py "" --rounds 2 --prof --batch 256 --hidden 32 2>&1 | grep -v amdgpu | tee $O/ab_c32.txt
python tools/exp/ab_step.py "" --rounds 2 --batch 256 --hidden 32 --hw 66 65 2>&1 | grep -v amdgpu | tee $O/ab_c32_66.txt
python tools/exp/gemm_shapes.py --splits 0 2>&1 | grep -v amdgpu | tee $O/gemm_shapes.txt
