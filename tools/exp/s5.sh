#!/bin/bash
# session 5: default route restored to round-3 arithmetic; act_pieces=3 with the cheap Nyquist split
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s5
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s5/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s5/pytest_gpu.log
for ap in 2 3 2 3; do CFD_ACT_PIECES=$ap python bench.py --only spectral 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin)['roofline_spectral_conv2d']; print('ap$ap', d['avg_us'], d['frac'])"; done
python tools/exp/ab_step.py "" "act_pieces=3" --rounds 3 > gpurun_out/s5/ab_pieces.txt 2>&1; cat gpurun_out/s5/ab_pieces.txt
CFD_ACT_PIECES=3 python tools/exp/spectral_err.py > gpurun_out/s5/err_act3.json 2> gpurun_out/s5/err.err
python tools/exp/spectral_err.py > gpurun_out/s5/err_act2.json 2>> gpurun_out/s5/err.err
