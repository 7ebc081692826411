#!/bin/bash
# EPIX 7 (one-pass raw + statistics epilogue): tests that cover the statistics kernels, then the benches
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/plain; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/race_screen.py > $O/race.log 2>&1; echo "race rc=$?"; tail -3 $O/race.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b6144.json 2> $O/bench_b6144.err; echo "bench rc=$?"; python3 -c "import json;d=json.load(open('$O/bench_b6144.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'],[ (k['kernel'][:20],k['ms_per_step']) for k in d['roofline']['by_kernel'][:3]])"
timeout -k 10 200 python3 bench.py --global-batch 768 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b768.json 2> $O/bench_b768.err; echo "bench768 rc=$?"; python3 -c "import json;d=json.load(open('$O/bench_b768.json'));print(d['ms_per_step'],d['value'])"
timeout -k 10 200 python3 tools/layer_breakdown.py 6144 > $O/layer_breakdown.txt 2>&1; head -14 $O/layer_breakdown.txt
