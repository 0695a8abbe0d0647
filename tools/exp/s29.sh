cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" old; do
  if [ -n "$v" ]; then export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_$v.so; else unset CFDBENCH_AMD_LIB; fi
  echo "== variant '$v'"
  for leg in unet resnet auto_deeponet_cnn; do
  python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print(d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:3]])"
  done
done
done
