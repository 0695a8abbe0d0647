#!/bin/bash
# GPU box: per-kernel busy counters of the bench step (which unit bounds each kernel) -> gpurun_out/<tag>/busy.txt
# usage: tools/pmc_step.sh <tag>
TAG=${1:-busy}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-rollout --no-extra"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc -o p -- $CMD > /dev/null 2> $OUT/pmc.err; echo "rc=$?"
python - <<PY > $OUT/busy.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':44s} {'n':>4s} {'cycles':>9s} {'mfma%':>6s} {'valu%':>6s} {'lds%':>6s} {'wait%':>6s} {'stall%':>6s}")
rows = []
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8          # summed over the 8 XCDs
    if cyc <= 0: continue
    simd = 1024.0
    rows.append((cyc * len(cs["GRBM_GUI_ACTIVE"]), k, len(cs["GRBM_GUI_ACTIVE"]), cyc,
                 m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd / cyc, 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / simd / cyc,
                 4 * m.get("SQ_ACTIVE_INST_LDS", 0) / simd / cyc,
                 m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
import json
js = {}
for _, k, n, cyc, a, b, c, d, e in sorted(rows, reverse=True)[:30]:
    print(f"{k[:44]:44s} {n:4d} {cyc:9.0f} {100*a:6.1f} {100*b:6.1f} {100*c:6.1f} {100*d:6.1f} {100*e:6.1f}")
    js[k] = dict(launches=n, cycles=round(cyc), mfma=round(100 * a, 1), valu=round(100 * b, 1), lds=round(100 * c, 1), wait=round(100 * d, 1),
                 stall=round(100 * e, 1))
json.dump(js, open("$OUT/step_busy.json", "w"), indent=1)
PY
cat $OUT/busy.txt
find $OUT -name "*.csv" -size +2M -delete
