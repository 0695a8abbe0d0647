#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes) per kernel of the U-Net and ResNet train steps -> gpurun_out/<tag>/
TAG=${1:-convtraffic}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in unet resnet; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${m}_$c -o p -- python $GRAFT_REPO_ROOT/tools/bench_$m.py > /dev/null 2> $OUT/${m}_$c.err; echo "$m $c rc=$?"
  done
done
python - <<PY
import csv, glob, json, collections
out = "$OUT"
for m in ("unet", "resnet"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/{m}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in agg.items():
        fetch = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [1])), 1)
        write = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [1])), 1)
        # units and the gfx950 correction as in tools/profile_step.sh (FETCH_SIZE counts 128-B requests as 64 B: doubled)
        res[k] = dict(fetch_kb_raw=round(fetch, 1), write_kb_raw=round(write, 1), traffic_bytes_per_launch=int((2 * fetch + write) * 1024),
                      launches=len(cs.get("FETCH_SIZE", [])))
    json.dump(res, open(out + f"/{m}_pmc_traffic.json", "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:10]:
        print(m, k[:56].ljust(56), v)
PY
find $OUT -name "*.csv" -size +2M -delete
