#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s9
cd /tmp && export TMPDIR=/tmp
for leg in deeponet auto_ffn auto_deeponet_cnn; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s9/t_$leg -o sp -- python $GRAFT_REPO_ROOT/bench.py --only $leg > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/s9/$leg.err
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/s9/t_$leg -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/s9/${leg}_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/s9/t_$leg
done
