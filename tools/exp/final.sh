#!/bin/bash
# final session of a round: GPU suite, full bench line, rocprofv3 stats + PMC traffic + busy counters of the headline step, PMC traffic of the legs
# usage: tools/exp/final.sh <tag>     -> gpurun_out/<tag>/...   (copy what is wanted to profiles/<tag>_*)
TAG=${1:-r04z}
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
bash tools/profile_step.sh $TAG/step > $OUT/profile_step.log 2>&1; tail -3 $OUT/profile_step.log
bash tools/pmc_step.sh $TAG/busy > $OUT/pmc_step.log 2>&1; tail -3 $OUT/pmc_step.log
for leg in spectral unet auto_deeponet; do
  timeout 900 bash tools/pmc_traffic.sh $TAG $leg python $GRAFT_REPO_ROOT/bench.py --only $leg > $OUT/pmc_$leg.log 2>&1; tail -2 $OUT/pmc_$leg.log
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/unet_trace -o u -- python $GRAFT_REPO_ROOT/bench.py --only unet > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$OUT/unet_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$OUT/unet_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/$OUT/unet_trace
python -c "import torch; print(torch.cuda.get_device_name(0), torch.version.hip)" > $GRAFT_REPO_ROOT/$OUT/env.log 2>&1; rocminfo 2>/dev/null | grep -m3 -i "gfx\|Compute Unit" >> $GRAFT_REPO_ROOT/$OUT/env.log
find $GRAFT_REPO_ROOT/$OUT -name "*.csv" -size +2M -delete
