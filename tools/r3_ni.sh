#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$PWD
O=$R/gpurun_out/ni; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_packed_roles_gpu.py tests/test_model_gpu.py tests/test_parallel_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b6144.json 2> $O/bench_b6144.err; echo "bench rc=$?"; python3 -c "import json;d=json.load(open('$O/bench_b6144.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
SR_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/trace.log; echo "trace rc=$?"
grep -h "node_init" $(find $O/trace -name "*kernel_stats.csv" | head -1); rm -rf $O/trace
