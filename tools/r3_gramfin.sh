#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/gramfin; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/race_screen.py > $O/race.log 2>&1; echo "race rc=$?"; tail -1 $O/race.log
for gb in 6144 768; do timeout -k 10 300 python3 bench.py --global-batch $gb --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_b$gb.json 2> $O/bench.err; python3 -c "import json;d=json.load(open('$O/bench_b$gb.json'));print($gb, d['ms_per_step'],d['value'])"; done
cd /tmp && export TMPDIR=/tmp
SR_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o out --output-format csv -- python3 $R/bench.py --global-batch 768 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/trace.log; echo "trace rc=$?"
grep -h "gram_reduce\|gram_project\|bn_finalize" $(find $O/trace -name "*kernel_stats.csv" | head -1) | cut -c1-200; rm -rf $O/trace
