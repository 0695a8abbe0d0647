#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/exp/r5_final.sh r5z > gpurun_out/r5z_final.log 2>&1; grep -v Warning gpurun_out/r5z_final.log | head -12 | cut -c1-200
bash tools/exp/r5_prof.sh r5r > gpurun_out/r5r_prof.log 2>&1; tail -3 gpurun_out/r5r_prof.log | cut -c1-160
