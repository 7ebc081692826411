mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/full4.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/full4.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/r4/bench3_b6144.json 2> gpurun_out/r4/bench3_b6144.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r4/bench3_b6144.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'])"
tail -3 gpurun_out/r4/bench3_b6144.err
timeout -k 10 200 python bench.py --global-batch 768 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c100-230
