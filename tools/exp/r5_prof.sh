#!/bin/bash
# round 5 profile set of the default (three-piece) route: kernel-trace stats + PMC traffic + busy counters of the bench step, counter traffic of
# the SpectralConv2d group and of the U-Net / Auto-DeepONet legs
cd $GRAFT_REPO_ROOT; T=${1:-r5p}
bash tools/profile_step.sh $T > gpurun_out/${T}_profile_step.log 2>&1; tail -12 gpurun_out/${T}_profile_step.log
bash tools/pmc_step.sh $T > gpurun_out/${T}_pmc_step.log 2>&1; head -14 gpurun_out/$T/busy.txt
bash tools/pmc_traffic.sh $T spectral python $GRAFT_REPO_ROOT/bench.py --only spectral > gpurun_out/${T}_spectral.log 2>&1; tail -8 gpurun_out/${T}_spectral.log
bash tools/pmc_traffic.sh $T unet python $GRAFT_REPO_ROOT/bench.py --only unet > gpurun_out/${T}_unet.log 2>&1; tail -6 gpurun_out/${T}_unet.log
bash tools/pmc_traffic.sh $T auto_deeponet python $GRAFT_REPO_ROOT/bench.py --only auto_deeponet > gpurun_out/${T}_adon.log 2>&1; tail -6 gpurun_out/${T}_adon.log
rocminfo | grep -m2 -i "marketing\|gfx" ; rocm-smi --showclocks 2>/dev/null | head -12
