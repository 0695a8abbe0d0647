#!/bin/bash
# GPU box: time bench legs under different settings, with the top kernel rows of each leg (the A/B sessions of round 4 were all of this form).
#   usage: tools/exp/legs_ab.sh "unet resnet" "" "CFD_CONV6_GRID=768" "CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_x.so"
#   first argument: legs (bench.py --only names); every further argument: one setting = space-separated ENV=VALUE pairs ("" = defaults)
cd $GRAFT_REPO_ROOT
LEGS=$1; shift
[ $# -eq 0 ] && set -- ""
for cfg in "$@"; do
  echo "== ${cfg:-defaults}"
  for leg in $LEGS; do
    env $cfg python bench.py --only $leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=list(json.load(sys.stdin).values())[0]
print('$leg', d['ms_per_step'], d.get('mode'), [(r['kernel'], r['launches_per_step'], r['us_per_step']) for r in d['kernels'][:6]])"
  done
done
