set -e
R=$(pwd); OUT=$R/gpurun_out/pmc_c128; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python3 $R/tools/pmc_c128.py > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; echo "pass $i failed"; }
done
cd $R
python3 - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_c128/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "c128" in k:
            agg[k[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for name, v in sorted(c.items()):
        print("   %-28s %16.0f per dispatch (%d)" % (name, sum(v) / len(v), len(v)))
PY
find gpurun_out/pmc_c128 -name "*.csv" -size +2M -delete
