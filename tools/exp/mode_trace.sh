#!/bin/bash
# Dev experiment (GPU box): rocprofv3 kernel-trace stats (true kernel durations) of tools/exp/mode_variants.py
# usage: tools/exp/mode_trace.sh <tag> [mode_variants args]
TAG=${1:-modes}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o modes -- python $GRAFT_REPO_ROOT/tools/exp/mode_variants.py "$@" > $OUT/trace_stdout.txt 2> $OUT/trace.err; echo "trace rc=$?"
grep -v amdgpu.ids $OUT/trace_stdout.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    print(f"{n[:70]:70s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
cp "$f" $OUT/modes_kernel_stats.csv
find $OUT/trace -type f -size +1M -delete
