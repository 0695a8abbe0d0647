#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s13; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["cpu_baseline"])
PY
bash tools/exp/r5_prof.sh r5q > $O/prof.log 2>&1; tail -5 $O/prof.log
