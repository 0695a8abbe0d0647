#!/bin/bash
# The determinism sessions of round 3 (root cause of k_head_fwd's wrong results beside another process: profiles/r03_det_root_cause.md)
# as ONE parametrised script: tools/det_sessions.sh <session 1..6> [tag].  Each arm is one gpurun call's worth of commands.
S=${1:-1}; shift || true
case $S in
1)
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

  ;;
2)
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

  ;;
3)
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

  ;;
4)
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

  ;;
5)
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

  ;;
6)
  set -u
  TAG=${1:-det6}; OUT=gpurun_out/$TAG; mkdir -p $OUT
  export TMPDIR=/tmp
  CFDBENCH_AMD_LIB=cfdbench_amd/_C/libcfdbench_amd_r2head.so REPS=3000 timeout 300 python tools/det_inproc.py 2>&1 | grep -v amdgpu.ids | tee $OUT/inproc_r2head.log
  CFDBENCH_AMD_LIB=cfdbench_amd/_C/libcfdbench_amd_fix1.so REPS=3000 timeout 300 python tools/det_inproc.py 2>&1 | grep -v amdgpu.ids | tee $OUT/inproc_fix1.log

  ;;
*) echo "usage: tools/det_sessions.sh <1..6> [tag]"; exit 1 ;;
esac
