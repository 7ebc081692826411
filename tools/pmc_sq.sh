#!/bin/bash
# SQ counter passes over one of the tools/pmc_*.py launch scripts (separate --pmc runs, nothing but the counters beside them; the
# program itself after `--`).  Summary: counter per dispatch for every kernel whose name matches PATTERN.
# usage: bash tools/pmc_sq.sh <out tag> <kernel name pattern (regex)> <script> [script args...]
set -e
R=$(pwd)
TAG=$1; PAT=$2; shift 2
SCRIPT=$(readlink -f "$1"); shift; set -- "$SCRIPT" "$@"
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
python3 "$@" > $OUT/times.txt 2>&1
cat $OUT/times.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python3 "$@" > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; echo "pass $i failed"; }
  echo "pass $i done"
done
cd $R
python3 - "$OUT" "$PAT" <<'PY'
import csv, collections, glob, re, sys
out, pat = sys.argv[1], re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat.search(k):
            agg[(k[:110], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    print(open(out + "/times.txt").read(), file=o)
    for (k, g), c in agg.items():
        print(k, "grid", g, file=o)
        for name, v in sorted(c.items()):
            print("   %-30s %16.0f per dispatch (%d dispatches)" % (name, sum(v) / len(v), len(v)), file=o)
print(open(out + "/summary.txt").read())
PY
find $OUT -name "*.csv" -size +2M -delete
