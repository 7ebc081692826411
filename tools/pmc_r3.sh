# SQ counter passes for the round-3 targets (separate --pmc runs, no tracing options beside them)
set -e
R=$(pwd)
OUT=$R/gpurun_out/pmc_r3
mkdir -p $OUT
python3 tools/pmc_r3.py all > $OUT/times.txt 2>&1
cat $OUT/times.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python3 $R/tools/pmc_r3.py all > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; echo "pass $i failed"; }
  echo "pass $i done"
done
cd $R
python3 - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_r3/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gram_kernel" in k or "conv_igemm" in k:
            agg[k[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_r3/summary.txt", "w") as out:
    for k, c in agg.items():
        print(k, file=out)
        for name, v in sorted(c.items()):
            print("   %-28s %16.0f per dispatch (%d dispatches)" % (name, sum(v) / len(v), len(v)), file=out)
print(open("gpurun_out/pmc_r3/summary.txt").read())
PY
find gpurun_out/pmc_r3 -name "*.csv" -size +2M -delete
