#!/bin/bash
# session 13: head templated on the piece count: default route unchanged?  exact route: errors, step, full suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s13
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s13/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s13/pytest_gpu.log
python tools/exp/ab_step.py "" "act_pieces=3" --rounds 3 > gpurun_out/s13/ab_pieces.txt 2>&1; cat gpurun_out/s13/ab_pieces.txt
CFD_ACT_PIECES=3 python tools/exp/spectral_err.py > gpurun_out/s13/err_act3.json 2> gpurun_out/s13/err.err; python -c "
import json; d=json.load(open('gpurun_out/s13/err_act3.json')); print({k: max(v.values()) for k,v in d.items() if k!='fno_model_B4_C20_L4'}); m=d['fno_model_B4_C20_L4']; print('model', {k:v for k,v in m.items() if k in ('preds','nmse_loss')}, 'max grad', max(v for k,v in m.items() if k.startswith('g:')))"
CFD_ACT_PIECES=3 python bench.py --no-cpu-baseline --no-rollout --no-extra 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('exact route step', d['ms_per_step']); [print('  ', k['kernel'], k['launches_per_step'], k['avg_us']) for k in d['kernels'][:6]]"
