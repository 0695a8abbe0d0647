#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6
cd /tmp && export TMPDIR=/tmp
for ap in 2 3; do
  CFD_ACT_PIECES=$ap timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s6/t$ap -o sp -- python $GRAFT_REPO_ROOT/bench.py --only spectral > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/s6/t$ap.err
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/s6/t$ap -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/s6/spectral_ap${ap}_kernel_stats.csv
  echo "== ap $ap"; head -8 $f | cut -c1-150
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/s6/t2 $GRAFT_REPO_ROOT/gpurun_out/s6/t3
