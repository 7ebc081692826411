set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/race_expand.py 20 2>&1 | grep -v amdgpu.ids; rc=$?
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/bench_expand.py 3 2 1 2>&1 | grep layer
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r2_tests4.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/r2_tests4.log
[ $rc -lt 124 ] || exit $rc
for ov in 0 1; do
SR_OVERLAP=$ov timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r2_bench_ov$ov.json 2> gpurun_out/r2_bench_ov$ov.err; echo "ov=$ov $(python3 -c "
import json
d=json.load(open('gpurun_out/r2_bench_ov$ov.json')); print(d['value'], d['ms_per_step'])")"
done
