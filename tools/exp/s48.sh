cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "linear or ffn or deeponet or DeepONet or golden or nonauto" 2>&1 | tail -3
for leg in auto_ffn auto_deeponet_cnn deeponet auto_deeponet; do
python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('$leg', d['ms_per_step'], [(r['kernel'], r['launches_per_step'], r['us_per_step']) for r in d['kernels'][:4]])"
done
