cd $GRAFT_REPO_ROOT
for v in "" lds110; do
  if [ -n "$v" ]; then export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_$v.so; else unset CFDBENCH_AMD_LIB; fi
  for leg in resnet auto_deeponet_cnn; do
  python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('variant $v $leg', d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:3]])"
  done
done
