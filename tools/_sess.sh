mkdir -p gpurun_out/r02f
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -s > gpurun_out/r02f/pytest.log 2>&1; grep -E "passed|failed|bf16 storage|rollout nMSE|preds nMSE" gpurun_out/r02f/pytest.log | tail -12
timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err; tail -3 gpurun_out/r02f/bench.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r02f/bench.json"))
for k in ("value","ms_per_step","rollout","rollout_66x65","rollout_66x65_bf16"):
    print(k, r.get(k))
PY
