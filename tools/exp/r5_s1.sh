#!/bin/bash
# round 5 session 1: baseline of the fp32-class route (act_pieces = 3) beside the two-piece route, per-kernel times of both
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s1; mkdir -p $O
python tools/exp/ab_step.py "act_pieces=3" "act_pieces=2" --rounds 3 --prof > $O/ab_ap3_prof.txt 2>&1
python tools/exp/ab_step.py "act_pieces=2" --rounds 1 --prof > $O/ab_ap2_prof.txt 2>&1
cat $O/ab_ap3_prof.txt $O/ab_ap2_prof.txt
python tools/kbench.py --tune act_pieces=3 > $O/kbench_ap3.txt 2>&1
python tools/kbench.py --batch 256 --hidden 32 --tune act_pieces=3 > $O/kbench_ap3_c32.txt 2>&1
rocminfo | grep -i -m3 "compute unit\|max clock" > $O/env.txt
