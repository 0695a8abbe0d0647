#!/bin/bash
# One gpurun call (round 3): is k_head_fwd's wrong-result-beside-another-process bug the private segment (register spills)?
#   tools/det_session.sh <tag>      -> gpurun_out/<tag>/*.log
# Arms (victim = head_fwd looping in one process, aggressor = a second process looping head_bwd on the same GPU):
#   A  shipped-r2 build        C = 20  (k_head_fwd<5,...>: 3 waves/SIMD target, 6 VGPRs spilled, reloaded in the tile loop)
#   B  shipped-r2 build        C = 8   (k_head_fwd<2,...>: same code, no spills)
#   C  -DCFD_HEAD_FWD_OCC=2    C = 20  (no spills)
#   D  OCC=2 + FORCE_SCRATCH   C = 20  (no spills, but one value parked in scratch in the prologue and re-read per tile)
#   E  tools/exp/scratch_cotenancy: a 40-line kernel that parks a signature in scratch, beside a foreign MFMA kernel
set -u
TAG=${1:-det}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPS=${REPS:-4000}
two() {  # name lib ch
    local name=$1 lib=$2 ch=$3
    CFDBENCH_AMD_LIB=$lib CH=$ch ONLY=head_bwd REPS=1000000 BATCHES=37 timeout 600 python tools/det_kernels.py > $OUT/${name}_aggr.log 2>&1 &
    local ap=$!
    sleep 8
    CFDBENCH_AMD_LIB=$lib CH=$ch ONLY=head_fwd REPS=$REPS BATCHES=4,37,256 timeout 500 python tools/det_kernels.py > $OUT/${name}_victim.log 2>&1
    kill $ap 2>/dev/null; wait $ap 2>/dev/null
    echo "== $name"; grep -c . $OUT/${name}_aggr.log; grep "^B=" $OUT/${name}_victim.log
}
L=cfdbench_amd/_C
two A_r2_c20 $L/libcfdbench_amd_r2head.so 20
two B_r2_c8 $L/libcfdbench_amd_r2head.so 8
two C_occ2_c20 $L/libcfdbench_amd_occ2.so 20
two D_occ2_scratch_c20 $L/libcfdbench_amd_occ2s.so 20
echo "== E scratch_cotenancy"
E=tools/exp/scratch_cotenancy_exp
timeout 120 $E victim 3000 > $OUT/E_alone.log 2>&1; cat $OUT/E_alone.log
timeout 300 $E aggr 100000 > $OUT/E_aggr.log 2>&1 &
ap=$!; sleep 3
timeout 200 $E victim 20000 > $OUT/E_victim.log 2>&1; cat $OUT/E_victim.log
timeout 200 $E novictim 20000 > $OUT/E_novictim.log 2>&1; cat $OUT/E_novictim.log
kill $ap 2>/dev/null; wait $ap 2>/dev/null
timeout 200 $E victim 20000 > $OUT/E_vv1.log 2>&1 &
vp=$!
timeout 200 $E victim 20000 > $OUT/E_vv2.log 2>&1; wait $vp; cat $OUT/E_vv1.log $OUT/E_vv2.log
