cd $GRAFT_REPO_ROOT
for cfg in "128 16" "512 16" "512 8" "512 4" "128 4" "128 8"; do
  set -- $cfg
  export CFD_EXP_SPLIT_TILES=$1 CFD_EXP_SPLIT_SLABS=$2
  for leg in auto_deeponet auto_edeeponet; do
  python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('tiles<$1 slabs>=$2 $leg', d['ms_per_step'], [(r['kernel'], r['launches_per_step'], r['us_per_step']) for r in d['kernels'] if r['kernel'] in ('k_gemm','k_splitk_reduce')])"
  done
done
