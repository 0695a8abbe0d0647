# GPU box: U-Net + ResNet legs with per-kernel rows
cd $GRAFT_REPO_ROOT
python bench.py --only unet 2>/dev/null | tail -1 > gpurun_out/s24_unet.json
python bench.py --only resnet 2>/dev/null | tail -1 > gpurun_out/s24_resnet.json
python - <<PY
import json
for n in ("unet", "resnet"):
    d = json.load(open("gpurun_out/s24_%s.json" % n))
    d = list(d.values())[0]
    print(n, d["ms_per_step"], d.get("eager_ms_per_step"))
    for r in d["kernels"]: print("   ", r)
PY
