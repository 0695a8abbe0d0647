cd $GRAFT_REPO_ROOT
export CFD_GEMM_TILE=64
for v in "" g1 g16 g32 g48 g17 g49; do
  if [ -n "$v" ]; then export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_$v.so; else unset CFDBENCH_AMD_LIB; fi
  python bench.py --only auto_ffn 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('variant $v', d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:2]])"
done
