#!/bin/bash
# full GPU check of a build: test suite, smoke, the default bench line (with the CPU baseline leg), env
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r5f}; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt; head -c 600 $O/bench.json
{ rocminfo | grep -m1 "Marketing Name.*MI\|gfx950"; rocm-smi --showproductname 2>/dev/null | head -8; nproc; python -c "import torch;print(torch.__version__, torch.version.hip)"; } > $O/env.log 2>&1
