cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or linear_act or deeponet_inner" 2>&1 | tail -1
for leg in deeponet auto_ffn auto_deeponet_cnn auto_deeponet auto_edeeponet; do
python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('$leg', d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:3]])"
done
