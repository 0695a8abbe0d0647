#!/bin/bash
# session 2 of round 4: full GPU suite, full bench line, PMC traffic of the spectral group and the other model legs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/s2/bench.json 2> gpurun_out/s2/bench.err; echo "bench rc=$?"
for leg in spectral unet auto_deeponet; do
  timeout 900 bash tools/pmc_traffic.sh s2 $leg python $GRAFT_REPO_ROOT/bench.py --only $leg
done
