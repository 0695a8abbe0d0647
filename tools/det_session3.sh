#!/bin/bash
# round 3, third determinism session: the diagnostic build of k_head_fwd (every broadcast LDS read of fc1 bias / fc2 weights checked for
# an exact zero, with hardware ids) as the victim beside a k_head_bwd aggressor process
set -u
TAG=${1:-det3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=cfdbench_amd/_C/libcfdbench_amd_r2head.so
D=cfdbench_amd/_C/libcfdbench_amd_dbg.so
CFDBENCH_AMD_LIB=$L CH=20 ONLY=head_bwd REPS=1000000 BATCHES=37 timeout 600 python tools/det_kernels.py > $OUT/aggr.log 2>&1 &
ap=$!
sleep 8
for b in 4 37 256; do
  CFDBENCH_AMD_LIB=$D CH=20 ONLY=head_fwd REPS=1000 BATCHES=$b DBG_FETCH=1 timeout 300 python tools/det_kernels.py > $OUT/victim_$b.log 2>&1
  grep -v "amdgpu.ids" $OUT/victim_$b.log | grep "cfd_debug_fetch\|kind=\|^B=" | head -90
done
kill $ap 2>/dev/null; wait $ap 2>/dev/null
