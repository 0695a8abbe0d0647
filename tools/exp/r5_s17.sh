#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -k "lifting_layer or fno or engine or whole_model" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python tools/exp/ab_step.py "" "stem_fuse=0" --rounds 3 --prof 2>&1 | grep -v amdgpu | tee $O/ab.txt | head -24
python tools/exp/ab_step.py "" "stem_fuse=0" --rounds 2 --batch 8 2>&1 | grep -v amdgpu | tee $O/ab_b8.txt | head -3
python tools/exp/ab_step.py "" "stem_fuse=0" --rounds 2 --batch 64 2>&1 | grep -v amdgpu | tee $O/ab_b64.txt | head -3
