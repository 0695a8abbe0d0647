#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5s11
bash tools/exp/graph_step_kernels.sh unet r5s11u 2>&1 | cut -c1-170 | tee gpurun_out/r5s11/unet_graph_step.txt
bash tools/exp/graph_step_kernels.sh auto_deeponet r5s11a 2>&1 | cut -c1-170 | tee gpurun_out/r5s11/adon_graph_step.txt
