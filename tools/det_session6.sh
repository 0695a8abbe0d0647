#!/bin/bash
set -u
TAG=${1:-det6}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CFDBENCH_AMD_LIB=cfdbench_amd/_C/libcfdbench_amd_r2head.so REPS=3000 timeout 300 python tools/det_inproc.py 2>&1 | grep -v amdgpu.ids | tee $OUT/inproc_r2head.log
CFDBENCH_AMD_LIB=cfdbench_amd/_C/libcfdbench_amd_fix1.so REPS=3000 timeout 300 python tools/det_inproc.py 2>&1 | grep -v amdgpu.ids | tee $OUT/inproc_fix1.log
