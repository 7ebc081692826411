#!/bin/bash
# After removing the compiler-inserted vmcnt drains (gemm <.,0>/<.,6>, c3d, fp8): GPU tests, headline bench, config 5, eval-mode backbone pass.
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/drain; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b6144.json 2> $O/bench_b6144.err; echo "bench rc=$?"; cat $O/bench_b6144.json
timeout -k 10 200 python3 bench.py --global-batch 768 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b768.json 2> $O/bench_b768.err; echo "bench768 rc=$?"; cat $O/bench_b768.json
timeout -k 10 300 python3 bench.py --fp8 --T 8 --global-batch 8192 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5_fp8.json 2> $O/bench_c5_fp8.err; echo "c5 rc=$?"; cat $O/bench_c5_fp8.json
timeout -k 10 200 python3 tools/eval_pass.py > $O/eval_pass.txt 2>&1; echo "eval rc=$?"; cat $O/eval_pass.txt
cd /tmp && export TMPDIR=/tmp
SR_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.log; echo "trace rc=$?"
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/trace
