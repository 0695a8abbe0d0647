cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "ffn or deeponet or DeepONet or edeeponet or auto_ffn or dense" 2>&1 | tail -3
for leg in auto_deeponet auto_edeeponet deeponet auto_deeponet_cnn; do
python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('$leg', d['ms_per_step'], [(r['kernel'], r['launches_per_step'], r['us_per_step']) for r in d['kernels'][:5]])"
done
