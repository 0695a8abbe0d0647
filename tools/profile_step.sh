#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of the bench step + two PMC passes (FETCH_SIZE / WRITE_SIZE) -> gpurun_out/<tag>/
# usage: tools/profile_step.sh <tag>
TAG=${1:-prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-rollout --no-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o step -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2> $OUT/pmc_fetch.err; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > /dev/null 2> $OUT/pmc_write.err; echo "write rc=$?"
python - <<PY
import csv, glob, json, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in agg.items():
    short = k.split("(")[0].replace("void ", "")
    fetch = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [1])), 1)
    write = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [1])), 1)
    # rocprofv3 units: KiB-ish (x1024?) -- FETCH_SIZE/WRITE_SIZE are reported in KB; gfx950 FETCH_SIZE counts 128-B requests as
    # 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM section): doubled here.
    res[short] = dict(fetch_kb_raw=round(fetch, 1), write_kb_raw=round(write, 1),
                      traffic_bytes=int((2 * fetch + write) * 1024), launches=len(cs.get("FETCH_SIZE", [])))
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes"])[:25]:
    print(f"{k[:60]:60s} {v}")
PY
for f in $(find $OUT/trace -name "*kernel_stats.csv" | head -1); do cp $f $OUT/step_kernel_stats.csv; head -25 $f | cut -c1-160; done
find $OUT -name "*.csv" -size +2M -delete
