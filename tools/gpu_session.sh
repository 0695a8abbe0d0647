#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel stats.  Everything is logged under gpurun_out/.
# usage: tools/gpu_session.sh [tag]
set -u
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env ==" > $OUT/env.log
(rocminfo | grep -E "Marketing Name|gfx" | head -6; nproc; lscpu | grep "Model name"; python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))") >> $OUT/env.log 2>&1
echo "== pytest gpu ==" 
timeout 500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== bench =="
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof kernel stats =="
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err); echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -30 $f; done
# keep only the small csv summaries
find $OUT/prof -type f ! -name "*stats*.csv" -size +2M -delete
