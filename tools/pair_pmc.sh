#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) and SQ counters of the fused pair kernel at batch 6144.
# usage: tools/pair_pmc.sh <outdir>
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/r5/pmc_pair}
mkdir -p $O
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/$n -o out --output-format csv -- python3 tools/pair_time.py 6144 time > $O/$n.log 2>&1 || echo "pass $n failed"
  f=$(find $O/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_kernel.py $f pair_kernel
done
