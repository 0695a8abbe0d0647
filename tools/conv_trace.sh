#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of the U-Net and ResNet train steps -> gpurun_out/<tag>/
TAG=${1:-convtrace}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in unet resnet; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o t -- python $GRAFT_REPO_ROOT/tools/bench_$m.py > $OUT/$m.json 2> $OUT/$m.err; echo "$m rc=$?"
  f=$(find $OUT/$m -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/${m}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${m}_kernel_stats.csv")))
for r in rows[:22]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), r['AverageNs'].rjust(12), r['Percentage'])
PY
done
