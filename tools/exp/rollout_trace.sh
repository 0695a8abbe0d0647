#!/bin/bash
# kernel trace of eager rollout steps + graph rates for a few configurations
set -u
TAG=${1:-ro_trace}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
for cfg in "64 64 20 64 200 f32" "66 65 32 64 200 f32" "66 65 32 64 200 bf16"; do
  name=$(echo $cfg | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$name -o t --output-format csv -- python /root/repo/tools/exp/rollout_trace.py $cfg > $OUT/$name.log 2>&1
  grep rollout $OUT/$name.log
  f=$(find $OUT/$name -name "t_kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"   {r['Name'][:90]:90s} n={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:7.1f} us  {float(r['Percentage']):5.1f}%")
PY
done
