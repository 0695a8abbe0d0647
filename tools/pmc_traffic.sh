#!/bin/bash
# GPU box: HBM traffic per kernel launch of an arbitrary command, as MI355X_MICROARCH.md prescribes: two separate rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; FETCH_SIZE doubled (gfx950 counts a 128-B request as 64 B for wide
# coalesced reads), both in KiB.   usage: tools/pmc_traffic.sh <tag> <name> <command...>   -> gpurun_out/<tag>/<name>_pmc_traffic.json
TAG=$1; NAME=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${NAME}_fetch -o p -- "$@" > $OUT/${NAME}_cmd.log 2> $OUT/${NAME}_fetch.err; echo "$NAME fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${NAME}_write -o p -- "$@" > /dev/null 2> $OUT/${NAME}_write.err; echo "$NAME write rc=$?"
python - <<PY
import csv, glob, json, collections
out, name = "$OUT", "$NAME"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/{name}_fetch/**/*counter_collection.csv", recursive=True) + glob.glob(f"{out}/{name}_write/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in agg.items():
    short = k.split("(")[0].replace("void ", "")
    nf, nw = len(cs.get("FETCH_SIZE", [])), len(cs.get("WRITE_SIZE", []))
    fetch = sum(cs.get("FETCH_SIZE", [0])) / max(nf, 1)
    write = sum(cs.get("WRITE_SIZE", [0])) / max(nw, 1)
    e = res.setdefault(short, dict(fetch_kb_raw=0.0, write_kb_raw=0.0, traffic_bytes=0, launches=0))
    # template instances that share a short name are merged launch-weighted
    tot = e["launches"] + nf
    if tot:
        e["fetch_kb_raw"] = round((e["fetch_kb_raw"] * e["launches"] + fetch * nf) / tot, 1)
        e["write_kb_raw"] = round((e["write_kb_raw"] * e["launches"] + write * nf) / tot, 1)
        e["launches"] = tot
        e["traffic_bytes"] = int((2 * e["fetch_kb_raw"] + e["write_kb_raw"]) * 1024)
res["_total_bytes_all_launches"] = int(sum(v["traffic_bytes"] * v["launches"] for v in res.values() if isinstance(v, dict)))
json.dump(res, open(f"{out}/{name}_pmc_traffic.json", "w"), indent=1)
for k, v in sorted(((k, v) for k, v in res.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:14]:
    print(f"{k[:70]:70s} {v}")
PY
rm -rf $OUT/${NAME}_fetch $OUT/${NAME}_write
