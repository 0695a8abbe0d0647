#!/bin/bash
cd $GRAFT_REPO_ROOT
for leg in unet auto_deeponet; do CFDBENCH_DP_ALWAYS_EXCHANGE=1 python bench.py --only $leg --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d['config']['flat_gradient_bytes'])"; done
