#!/bin/bash
# Two PROCESSES on one GPU, each repeating every FNO-path entry point REPS times (default 10000) at two batch sizes, outputs compared
# bitwise with the first launch (tools/det_kernels.py).  VERDICT r2 "done" criterion: 0 differences in >= 10 k launches for all 17
# entry points.      tools/det_sweep.sh <tag> [reps]   -> gpurun_out/<tag>/det_sweep.txt
set -u
TAG=${1:-sweep}; REPS=${2:-10000}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPS=$REPS BATCHES=4,37 timeout 2400 python tools/det_kernels.py > $OUT/p1.log 2>&1 &
p1=$!
REPS=$REPS BATCHES=37,4 timeout 2400 python tools/det_kernels.py > $OUT/p2.log 2>&1
wait $p1
{ echo "two concurrent processes x 2 batch sizes x 17 entry points x $REPS launches (tools/det_sweep.sh), library $(python -c 'from cfdbench_amd import _lib; print(_lib.lib_path().name)')"
  echo "process 1:"; grep "^B=" $OUT/p1.log; echo "process 2:"; grep "^B=" $OUT/p2.log
  echo "series: $(cat $OUT/p1.log $OUT/p2.log | grep -c '^B=')  ok: $(cat $OUT/p1.log $OUT/p2.log | grep '^B=' | grep -c ' ok$')"; } | tee $OUT/det_sweep.txt
