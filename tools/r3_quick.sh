#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/quick; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_production_shapes_gpu.py tests/test_kernels_gpu.py tests/test_blocks_teacher_forced_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/race_screen.py > $O/race.log 2>&1; echo "race rc=$?"; tail -1 $O/race.log
for i in 1 2; do timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b6144_$i.json 2> $O/bench_b6144.err; echo "bench rc=$?"; python3 -c "import json;d=json.load(open('$O/bench_b6144_$i.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'],[ (k['kernel'][:20],k['ms_per_step']) for k in d['roofline']['by_kernel'][:3]])"; done
timeout -k 10 200 python3 tools/layer_breakdown.py 6144 > $O/layer_breakdown.txt 2>&1; head -8 $O/layer_breakdown.txt
