#!/bin/bash
# GPU box: busy counters per kernel for an arbitrary command.  usage: tools/pmc_cmd.sh <tag> <command...>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc -o p -- "$@" > $OUT/cmd.log 2> $OUT/pmc.err; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -o p -- "$@" > /dev/null 2>> $OUT/pmc.err; echo "rc=$?"
python - <<PY
import csv, glob, collections
for d in ("$OUT/pmc", "$OUT/pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if not k.startswith("k_"): continue
        print(k[:50], len(next(iter(cs.values()))), {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
find $OUT -name "*.csv" -size +2M -delete
