#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s17
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s17/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s17/pytest_gpu.log
for leg in auto_deeponet deeponet auto_ffn auto_deeponet_cnn auto_edeeponet; do for g in 0 1; do CFD_GEMM6=$g python bench.py --only $leg 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=list(d)[0]; v=d[k]
print('gemm6=$g', k, v.get('ms_per_step'), v.get('mode'), [ (r['kernel'], r['launches_per_step'], r['us_per_step']) for r in v.get('kernels', [])[:3]])
"; done; done
