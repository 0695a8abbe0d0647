#!/bin/bash
# round 3, second determinism session: dumps of failing head_fwd outputs for offline analysis + the C = 8 victim beside a C = 20 aggressor
set -u
TAG=${1:-det2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=cfdbench_amd/_C/libcfdbench_amd_r2head.so
two() {  # name victim_ch aggr_ch reps batches [dump]
    local name=$1 vch=$2 ach=$3 reps=$4 batches=$5
    CFDBENCH_AMD_LIB=$L CH=$ach ONLY=head_bwd REPS=1000000 BATCHES=37 timeout 600 python tools/det_kernels.py > $OUT/${name}_aggr.log 2>&1 &
    local ap=$!
    sleep 8
    CFDBENCH_AMD_LIB=$L CH=$vch ONLY=head_fwd REPS=$reps BATCHES=$batches DUMP_DIR=$OUT DUMP_MAX=${6:-0} timeout 500 python tools/det_kernels.py > $OUT/${name}_victim.log 2>&1
    kill $ap 2>/dev/null; wait $ap 2>/dev/null
    echo "== $name"; grep "^B=" $OUT/${name}_victim.log
}
two A_dump 20 20 3000 4 12
two B_c8_beside_c20 8 20 6000 4,37,256
two F_c20_beside_c8 20 8 3000 4,37,256
