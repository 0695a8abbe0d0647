#!/bin/bash
# per-launch durations of one bench leg's kernels, grouped by (kernel, grid): tools/exp/leg_trace.sh <leg> [tag]
LEG=${1:-unet}; OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-leg_$LEG}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o u -- python $GRAFT_REPO_ROOT/bench.py --only $LEG > $OUT/cmd.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = (n.split("(")[0].replace("void ", "")[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    v = sorted(v)
    print(f"{k[0]:60s} grid {k[1]:>8s} {k[2]:>4s} {k[3]:>3s}  n={len(v):5d}  med {v[len(v)//2]:8.1f} us  {100*sum(v)/tot:5.1f} %")
PY
rm -rf $OUT/t
