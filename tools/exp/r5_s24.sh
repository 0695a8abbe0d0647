#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_harness.py -q -k "loss or unet or deeponet or ffn or resnet or graph" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
for leg in auto_deeponet auto_edeeponet unet; do python bench.py --only $leg 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print(k, v['ms_per_step'], v.get('eager_ms_per_step'))"; done
