set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r2_tests8.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r2_tests8.log; grep -E "^FAILED" gpurun_out/r2_tests8.log
[ $rc -lt 124 ] || exit $rc
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench9.json 2> gpurun_out/r2_bench9.err; python3 -c "
import json
d=json.load(open('gpurun_out/r2_bench9.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
