cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "512 2" "512 1" "768 1" "1024 1" "256 2"; do
  set -- $cfg
  echo "== grid $1 mul $2"
  CFD_CONV6_GRID=$1 CFD_CONV6_WGRAD_MUL=$2 python tools/bench_unet.py --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unet', d['ms_per_step'], {k:v for k,v in d['hip_kernel_us_per_step'].items() if 'conv' in k or 'part' in k})"
  CFD_CONV6_GRID=$1 CFD_CONV6_WGRAD_MUL=$2 python tools/bench_resnet.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resnet', d['ms_per_step'], {k:v for k,v in d['hip_kernel_us_per_step'].items() if 'conv' in k or 'part' in k})"
done
