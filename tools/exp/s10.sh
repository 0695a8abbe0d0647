#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s10
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s10/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s10/pytest_gpu.log
for leg in deeponet auto_ffn auto_deeponet_cnn auto_edeeponet; do python bench.py --only $leg 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=list(d)[0]; v=d[k]
print(k, v.get('ms_per_step'), v.get('mode'), v.get('eager_ms_per_step'), v.get('error'))
for r in v.get('kernels', [])[:6]: print('    ', r)
"; done
