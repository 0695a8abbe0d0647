#!/bin/bash
# round 3, fourth determinism session: (1) k_head_fwd with the fc2 accumulation on plain packed FMAs (no operand-select broadcast) beside the
# k_head_bwd aggressor; (2) the packed-FMA operand-select forms in isolation (tools/exp/pkfma_cotenancy) beside the same aggressor
set -u
TAG=${1:-det4}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=cfdbench_amd/_C/libcfdbench_amd_r2head.so
F=cfdbench_amd/_C/libcfdbench_amd_fix1.so
CFDBENCH_AMD_LIB=$L CH=20 ONLY=head_bwd REPS=1000000 BATCHES=37 timeout 600 python tools/det_kernels.py > $OUT/aggr.log 2>&1 &
ap=$!
sleep 8
echo "== fixed k_head_fwd beside the aggressor"
CFDBENCH_AMD_LIB=$F CH=20 ONLY=head_fwd REPS=6000 BATCHES=4,37,256 timeout 400 python tools/det_kernels.py > $OUT/victim_fix1.log 2>&1
grep "^B=" $OUT/victim_fix1.log
echo "== round-2 k_head_fwd (control)"
CFDBENCH_AMD_LIB=$L CH=20 ONLY=head_fwd REPS=1000 BATCHES=4,256 timeout 300 python tools/det_kernels.py > $OUT/victim_r2.log 2>&1
grep "^B=" $OUT/victim_r2.log
echo "== packed-FMA forms beside the aggressor"
timeout 200 tools/exp/pkfma_cotenancy_exp 3000 1024 16 | tee $OUT/pkfma_beside.log
kill $ap 2>/dev/null; wait $ap 2>/dev/null
echo "== packed-FMA forms alone"
timeout 200 tools/exp/pkfma_cotenancy_exp 3000 1024 16 | tee $OUT/pkfma_alone.log
