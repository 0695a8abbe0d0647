#!/bin/bash
# session 3: three-piece activation operands -- errors vs fp64, GPU suite, step / spectral-group timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s3
python tools/exp/spectral_err.py > gpurun_out/s3/err_act3.json 2> gpurun_out/s3/err.err; echo "err rc=$?"; cat gpurun_out/s3/err_act3.json
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s3/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s3/pytest_gpu.log
python bench.py --no-cpu-baseline --no-rollout --no-extra > gpurun_out/s3/bench_quick.json 2> gpurun_out/s3/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/s3/bench_quick.json"))
print(d["ms_per_step"], d["value"], d["roofline_spectral_conv2d"]["avg_us"], d["roofline_spectral_conv2d"]["frac"])
for k in d["kernels"]: print(k["kernel"], k["launches_per_step"], k["avg_us"])
PY
