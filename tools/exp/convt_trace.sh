#!/bin/bash
# rocprofv3 kernel trace of a few eager U-Net steps: true durations of the ConvTranspose2d kernels, per launch (level)
set -u
TAG=${1:-convt_trace}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
for mf in 1 0; do
  CFD_CONVT_MFMA=$mf timeout 300 rocprofv3 --kernel-trace -d $OUT/t$mf -o t --output-format csv -- python /root/repo/tools/bench_unet.py --steps 4 > /root/repo/$OUT/run$mf.log 2>&1
  f=$(find $OUT/t$mf /root/repo/$OUT/t$mf -name "t_kernel_trace.csv" 2>/dev/null | head -1)
  echo "== mfma=$mf trace: $f"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "convt" in n or "k_conv_wgrad<true>" in n or "k_part_reduce" in n:
        g = (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
        d[(n[:60], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = v[len(v) // 2:]
    print(f"{k[0]:60s} grid={k[1]} n={len(v):3d} avg={sum(v)/len(v):8.1f} us")
PY
done
