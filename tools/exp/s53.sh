cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "zero_padding or cnn or conv2d" 2>&1 | tail -3
for i in 1 2; do
python bench.py --only auto_deeponet_cnn 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('cnn', d['ms_per_step'], [(r['kernel'], r['launches_per_step'], r['us_per_step']) for r in d['kernels'][:6]])"
done
