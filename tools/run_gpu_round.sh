set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r2_tests5.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r2_tests5.log
[ $rc -lt 124 ] || exit $rc
timeout -k 10 300 python tools/bench_layers.py 3 stem > gpurun_out/layers_now.txt 2>&1; grep -v amdgpu gpurun_out/layers_now.txt
for gb in 6144 3072 1536 768; do
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --global-batch $gb --no-cpu-baseline --no-roofline > gpurun_out/r2_bench_b$gb.json 2> gpurun_out/r2_bench_b$gb.err; echo "B=$gb $(python3 -c "
import json
d=json.load(open('gpurun_out/r2_bench_b$gb.json')); print(d['value'], d['ms_per_step'])")"
done
