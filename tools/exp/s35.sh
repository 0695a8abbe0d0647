cd $GRAFT_REPO_ROOT
for g in 512 768 1024; do
  export CFD_CONV6_GRID=$g
  echo "== CFD_CONV6_GRID=$g"
  for leg in unet resnet; do
  python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('$leg', d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:3]])"
  done
done
