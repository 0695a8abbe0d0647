#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_harness.py -q -x -k "graph_option_on_two_ranks" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log | cut -c1-250
