cd $GRAFT_REPO_ROOT
export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_c6wide.so
python tools/exp/c6_diag.py 128 12 12 64 3 fwd 2>&1 | grep -v amdgpu
python tools/exp/c6_diag.py 128 24 12 64 3 dgrad 2>&1 | grep -v amdgpu
python tools/exp/c6_diag.py 32 16 64 64 7 fwd 2>&1 | grep -v amdgpu | head -5
python bench.py --only unet 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print(d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:3]])"
python bench.py --only resnet 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print(d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:3]])"
