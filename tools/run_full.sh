#!/bin/bash
# The GPU suite, the default bench line and the 8-GPU share on the box (what the driver runs at round end, in one call): logs under gpurun_out/.
mkdir -p gpurun_out/run
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/run/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/run/tests.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/run/bench_b6144.json 2> gpurun_out/run/bench_b6144.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/run/bench_b6144.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout -k 10 200 python bench.py --global-batch 768 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('b768', d['ms_per_step'])"
