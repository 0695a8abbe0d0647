#!/bin/bash
# A/B of library variants (tools/build_variant.sh): the train step and its per-kernel HIP-event times, one process per variant, two passes
#   tools/exp/lib_ab.sh OUTDIR tag1 tag2 ...      ("base" = the regular build)
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; shift; mkdir -p $O
for pass in 1 2; do
for tag in "$@"; do
  if [ "$tag" = base ]; then unset CFDBENCH_AMD_LIB; else export CFDBENCH_AMD_LIB=$GRAFT_REPO_ROOT/cfdbench_amd/_C/libcfdbench_amd_$tag.so; fi
  python tools/exp/ab_step.py "" --rounds 2 --prof ${AB_ARGS} > $O/ab_${tag}_$pass.txt 2>&1
  echo "== $tag pass $pass"; grep -v amdgpu.ids $O/ab_${tag}_$pass.txt | head -${AB_LINES:-5}
done; done
