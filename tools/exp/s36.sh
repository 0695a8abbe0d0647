cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s36; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o p -- python $GRAFT_REPO_ROOT/bench.py --only auto_deeponet > $O/log.txt 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:24]: print("%9.1f us avg  %6s calls  %5.1f %%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["Percentage"]), r["Name"][:110]))
PY
g=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$g")))
agg = collections.defaultdict(list)
for r in rows:
    if "k_gemm" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()): print(k, len(v), "avg %.1f us" % (sum(v) / len(v)))
PY
find $O -name "*.csv" -size +1M -delete
