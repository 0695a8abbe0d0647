cd $GRAFT_REPO_ROOT
export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_c6diag.so
python tools/exp/c6_diag.py 128 12 12 64 3 fwd 2>&1 | grep -v amdgpu
python tools/exp/c6_diag.py 128 24 12 64 3 dgrad 2>&1 | grep -v amdgpu
python tools/exp/c6_diag.py 128 24 24 32 3 dgrad 2>&1 | grep -v amdgpu
python tools/exp/c6_diag.py 128 24 12 64 3 fwd 2>&1 | grep -v amdgpu
python tools/exp/c6_diag.py 32 16 64 64 7 fwd 2>&1 | grep -v amdgpu
