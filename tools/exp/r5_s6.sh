#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s6; mkdir -p $O
python tools/exp/gemm_shapes.py --splits 0,1,2,4,8,16,32,64,128 2>&1 | grep -v amdgpu | tee $O/gemm_shapes.txt
