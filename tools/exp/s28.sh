cd $GRAFT_REPO_ROOT
for v in "" dA dB; do
  if [ -n "$v" ]; then export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_$v.so; else unset CFDBENCH_AMD_LIB; fi
  echo "== variant '$v'"
  python bench.py --only unet 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print(d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:3]])"
done
for v in dAs dBs; do
  export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_$v.so
  echo "== variant '$v'"
  python tools/exp/c6_diag.py 128 12 12 64 3 fwd 2>&1 | grep -v amdgpu
done
