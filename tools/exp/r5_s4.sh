#!/bin/bash
# DVFS probe of the head: same instruction stream on random and on all-zero data
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s4; mkdir -p $O
for i in 1 2; do
python tools/kbench.py --only head_train,head_fwd,block_fwd_act,chan_wgrad_act,dft_act --reps 50 2>&1 | grep -v amdgpu | tee -a $O/kbench_rand.txt
python tools/kbench.py --only head_train,head_fwd,block_fwd_act,chan_wgrad_act,dft_act --reps 50 --zeros 2>&1 | grep -v amdgpu | tee -a $O/kbench_zeros.txt
python tools/kbench.py --only head_train,head_fwd --reps 50 --tune act_pieces=2 2>&1 | grep -v amdgpu | tee -a $O/kbench_ap2.txt
python tools/kbench.py --only head_train,head_fwd --reps 50 --tune act_pieces=2 --zeros 2>&1 | grep -v amdgpu | tee -a $O/kbench_ap2_zeros.txt
done
