set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/race_expand.py 16 2>&1 | grep -v amdgpu.ids; rc=$?
[ $rc -lt 124 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_production_shapes_gpu.py -m gpu -q --timeout 300 -k "expand or conv1x1_256_1024" > gpurun_out/r2_tests3.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r2_tests3.log
: > gpurun_out/expand_ab.txt
for rep in 1 2; do
for cfg in "ws:SR_NO_WS=0" "v1:SR_NO_WS=1"; do
  tag=${cfg%%:*}; kv=${cfg#*:}
  env $kv REP=8 timeout -k 10 120 python tools/bench_expand.py 3 2 1 2>&1 | grep "res+relu" | sed "s/^/$tag /" >> gpurun_out/expand_ab.txt
done; done
sort gpurun_out/expand_ab.txt
