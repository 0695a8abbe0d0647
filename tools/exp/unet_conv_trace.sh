#!/bin/bash
# per-launch durations of the conv kernels of one replayed U-Net step (forward / input gradient / weight gradient per layer shape)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-unet_trace}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o u -- python $GRAFT_REPO_ROOT/bench.py --only unet > $OUT/cmd.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("k_conv6", "k_fold", "k_bn_", "k_conv1", "k_convt6", "k_splitk")): continue
    key = (n.split("(")[0].replace("void ", "")[:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("LDS_Block_Size", ""))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k[0]:44s} grid {k[1]:>7s} {k[2]:>3s} {k[3]:>3s} lds {k[4]:>6s}  n={len(v):4d}  med {v[len(v)//2]:7.1f} us  total {sum(v)/1e3:7.2f} ms")
PY
rm -rf $OUT/t
