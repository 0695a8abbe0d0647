#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in -1 4 2 8; do echo "== mix_nwv=$v"; python tools/kbench.py --tune mix_nwv=$v --only mix,mix_adj,spectral_fwd,spectral_bwd --reps 50 2>&1 | grep -v "amdgpu\|^#"; done
