#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s18; mkdir -p $O
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_gpu_dp.py -q -x > $O/pytest_dp_$i.log 2>&1; echo "dp run $i rc=$?"; tail -1 $O/pytest_dp_$i.log; done
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_harness.py -q -k "graph or rollout" > $O/pytest_graph.log 2>&1; echo "graph rc=$?"; tail -1 $O/pytest_graph.log
