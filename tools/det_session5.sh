#!/bin/bash
# round 3: the operand-select forms of the packed fp32 instructions in isolation beside the k_head_bwd aggressor process, and beside OTHER aggressors
set -u
TAG=${1:-det5}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=cfdbench_amd/_C/libcfdbench_amd_r2head.so
for aggr in head_bwd head_train block_fwd_act dft mix_adj_wgrad chan_wgrad_act; do
  CFDBENCH_AMD_LIB=$L CH=20 ONLY=$aggr REPS=1000000 BATCHES=37 timeout 300 python tools/det_kernels.py > $OUT/aggr_$aggr.log 2>&1 &
  ap=$!
  sleep 7
  echo "== packed fp32 operand-select forms beside a foreign process looping $aggr"
  timeout 100 tools/exp/pkfma_cotenancy_exp 1000 1024 8 | tee $OUT/pkfma_beside_$aggr.log | grep -v " 0 0 0 0   hi: 0 0 0 0"
  kill $ap 2>/dev/null; wait $ap 2>/dev/null
done
