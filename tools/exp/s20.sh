#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "ffn or deeponet or dense or linear" 2>&1 | tail -2
for leg in auto_deeponet auto_edeeponet deeponet; do python bench.py --only $leg 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=list(d)[0]; v=d[k]
print(k, v.get('ms_per_step'), v.get('mode'), [ (r['kernel'], r['launches_per_step'], r['us_per_step']) for r in v.get('kernels', [])[:5]])
"; done
bash tools/pmc_cmd.sh s20 python $GRAFT_REPO_ROOT/bench.py --only auto_deeponet 2>&1 | grep "k_ffn_stack_bwd_chain"
