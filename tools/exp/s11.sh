#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s11
for leg in deeponet auto_ffn; do python bench.py --only $leg 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=list(d)[0]; v=d[k]
print(k, v.get('ms_per_step'), v.get('mode'), v.get('eager_ms_per_step'), v.get('error'))
for r in v.get('kernels', [])[:4]: print('    ', r)
"; done
CFD_CONV1_MFMA=1 timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s11/pytest_gpu_conv1.log 2>&1; echo "pytest(conv1_mfma=1) rc=$?"; tail -3 gpurun_out/s11/pytest_gpu_conv1.log
for v in 0 1 0 1; do CFD_CONV1_MFMA=$v python bench.py --only unet 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['unet_cfg2']; print('conv1_mfma=$v', d['ms_per_step'], d['mode'])"; done
for v in 0 1; do CFD_CONV1_MFMA=$v python bench.py --only resnet 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['resnet_b32']; print('resnet conv1_mfma=$v', d['ms_per_step'], d['mode'])"; done
