#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s20; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "lifting_layer" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python tools/exp/ab_step.py "" "stem_fuse=2" --rounds 3 --prof --hw 66 65 2>&1 | grep -v amdgpu | tee $O/ab_66.txt | head -22
