# usage: graph_step_kernels.sh LEG TAG  -> per-kernel calls and time of ONE replayed step (difference of a 10- and a 40-replay run)
LEG=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
for n in 10 40; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$n -o p -- python $GRAFT_REPO_ROOT/tools/exp/graph_only.py $LEG $n > $O/n$n.log 2>&1
done
python - <<PY
import csv, glob
def load(n):
    f = glob.glob("$O/n%d/**/p_kernel_stats.csv" % n, recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(10), load(40)
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0.0))
    if c > c0: rows.append((k, (c - c0) / 30, (t - t0) / 30 / 1e3))
rows.sort(key=lambda r: -r[2])
print("total us/step", round(sum(r[2] for r in rows), 1), "launches/step", sum(r[1] for r in rows))
for k, c, t in rows: print("%8.1f us %6.1f x  %s" % (t, c, k[:130]))
PY
grep replay $O/*.log
find $O -name "*.csv" -size +1M -delete
