cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "broadcast or rowdot or nonauto or deeponet" 2>&1 | tail -2
for i in 1 2; do
python bench.py --only deeponet 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('deeponet', d['ms_per_step'], [(r['kernel'], r['launches_per_step'], r['us_per_step']) for r in d['kernels'][:8]])"
done
