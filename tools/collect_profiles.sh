#!/bin/bash
# Collects the round's judged evidence on the GPU box into gpurun_out/prof_<tag>/ (copy the summaries into profiles/<tag>/):
#   kernel trace of the bench command, separate --pmc FETCH_SIZE / WRITE_SIZE passes (HBM bytes of the conv kernels), the bench JSON
#   under the profiler and without it, and the per-shape table of one backbone pass.
# usage: bash tools/collect_profiles.sh r02
set -o pipefail
tag=${1:-r04}
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
# per-kernel durations must be the kernel's own: the two backbones share one stream here (SR_OVERLAP=0), as in bench.py's roofline leg
SR_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o out --output-format csv -- $BENCH > $O/bench_under_rocprof.json 2> $O/trace.log; rc=$?
echo "trace rc=$rc"; [ $rc -lt 124 ] || exit $rc
SR_OVERLAP=0 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o out --output-format csv -- $BENCH --no-roofline > /dev/null 2> $O/fetch.log; rc=$?
echo "fetch rc=$rc"; [ $rc -lt 124 ] || exit $rc
SR_OVERLAP=0 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -o out --output-format csv -- $BENCH --no-roofline > /dev/null 2> $O/write.log; rc=$?
echo "write rc=$rc"; [ $rc -lt 124 ] || exit $rc
cd $R
python3 tools/pmc_summarize.py $(find $O/trace -name "*kernel_trace.csv" | head -1) $(find $O/fetch -name "*counter_collection.csv" | head -1) \
        $(find $O/write -name "*counter_collection.csv" | head -1) $O b6144
cp $(find $O/trace -name "*agent_info.csv" | head -1) $O/agent_info.csv 2>/dev/null
timeout -k 10 500 python3 bench.py --steps 10 --warmup 3 --cpu-batch2 256 > $O/bench_b6144.json 2> $O/bench_b6144.err; echo "bench rc=$?"
timeout -k 10 300 python3 tools/layer_breakdown.py 6144 > $O/layer_breakdown_b6144.txt 2>&1
# the other per-GPU shares of the global batch (strong scaling: 2 / 4 / 8 ranks), config 5 (fp8 3x3 convolutions, T = 8, batch 8192) and its
# bf16 twin, the CPU baseline at batch 256 (BASELINE.md plan), a kernel trace at the 8-GPU share
for gb in 3072 1536 768; do
  timeout -k 10 200 python3 bench.py --global-batch $gb --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b$gb.json 2> $O/bench_b$gb.err; echo "bench $gb rc=$?"
done
timeout -k 10 300 python3 bench.py --fp8 --T 8 --global-batch 8192 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5_fp8.json 2> $O/bench_c5_fp8.err; echo "c5 fp8 rc=$?"
timeout -k 10 300 python3 bench.py --T 8 --global-batch 8192 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err; echo "c5 bf16 rc=$?"
if [ "$CPU256" = "1" ]; then
  timeout -k 10 500 python3 bench.py --steps 3 --warmup 1 --no-roofline --cpu-batch 256 --cpu-steps 3 > $O/bench_cpu256.json 2> $O/bench_cpu256.err; echo "cpu256 rc=$?"
fi
# keep the merged output small: the raw traces are large
rm -rf $O/trace 2>/dev/null
find $O -name "*.csv" -size +20M -delete
ls -la $O
