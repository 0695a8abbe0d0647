#!/bin/bash
# round 6, session 1: the matrix-pipe mode kernels (modes.hip) -- parity on the GPU, kernel A/B against the VALU kernels, chunk sweep, step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "mix or spectral_fwd" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
for t in "mode_mfma=0" "mode_mfma=1" "mode_mfma=1,mode_bc=16" "mode_mfma=1,mode_bc=24" "mode_mfma=1,mode_bc=40" "mode_mfma=1,mode_bc=37"; do
  echo "== $t"; python tools/kbench.py --only mix,mix_adj,mix_adj_wgrad --reps 50 --tune $t 2>/dev/null | grep -v amdgpu.ids | tail -4
done
python tools/exp/ab_step.py "" "mode_mfma=0" --rounds 3 --prof 2>&1 | grep -v amdgpu.ids | head -40
