#!/bin/bash
# round 5 session 2: GPU test suite on the new default (three-piece activations), new DP / parity tests, smoke, quick bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; head -c 1500 $O/bench.json
