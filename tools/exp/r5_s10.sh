#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -k "ffn or stack or deeponet or dense or linear or edeeponet" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for leg in auto_deeponet auto_edeeponet; do python bench.py --only $leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]
print(k, v['ms_per_step'], v.get('eager_ms_per_step')); [print('   ',r) for r in v['kernels']]"; done
bash tools/pmc_traffic.sh r5s10 auto_deeponet python $GRAFT_REPO_ROOT/bench.py --only auto_deeponet > $O/adon_traffic.log 2>&1; grep "k_ffn_stack" $O/adon_traffic.log
