cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "both_block_tiles" 2>&1 | tail -1
for t in 64 8; do
  export CFD_GEMM_TILE=$t
  echo "== CFD_GEMM_TILE=$t"
  for leg in deeponet auto_ffn auto_deeponet_cnn; do
  python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('$leg', d['ms_per_step'], [(r['kernel'], r['us_per_step']) for r in d['kernels'][:2]])"
  done
done
