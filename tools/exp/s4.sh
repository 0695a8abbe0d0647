#!/bin/bash
# session 4: act_pieces 2 vs 3 (templated build): step A/B, spectral group per route, GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s4
python tools/exp/ab_step.py "" "act_pieces=3" --rounds 3 > gpurun_out/s4/ab_pieces.txt 2>&1; cat gpurun_out/s4/ab_pieces.txt
for ap in 2 3; do CFD_ACT_PIECES=$ap python bench.py --only spectral > gpurun_out/s4/spectral_ap$ap.json 2>/dev/null; cat gpurun_out/s4/spectral_ap$ap.json; echo; done
CFD_ACT_PIECES=3 python tools/exp/spectral_err.py > gpurun_out/s4/err_act3.json 2> gpurun_out/s4/err.err; echo "err rc=$?"
python tools/exp/spectral_err.py > gpurun_out/s4/err_act2.json 2>> gpurun_out/s4/err.err; echo "err rc=$?"
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s4/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s4/pytest_gpu.log
