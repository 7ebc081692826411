#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/loss; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b6144_$i.json 2> $O/bench_b6144.err; echo "bench rc=$?"; python3 -c "import json;d=json.load(open('$O/bench_b6144_$i.json'));print(d['ms_per_step'],d['value'],d['config']['final_loss'])"; done
