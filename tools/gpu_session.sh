#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel stats + PMC passes.  Everything is logged under gpurun_out/<tag>/.
# usage: tools/gpu_session.sh [tag] [pytest-args...]
set -u
TAG=${1:-r2}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env ==" > $OUT/env.log
(rocminfo | grep -E "Marketing Name|gfx" | head -6; nproc; lscpu | grep "Model name"; python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))") >> $OUT/env.log 2>&1
echo "== pytest gpu =="
timeout 1500 python -m pytest tests -m gpu -q --durations=15 "$@" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== bench =="
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
