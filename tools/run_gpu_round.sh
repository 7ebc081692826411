set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_full_configs_gpu.py -m gpu -q -s --timeout 500 > gpurun_out/r2_tests2.log 2>&1; rc=$?
echo "pytest rc=$rc"; grep -E "config|passed|failed|Error" gpurun_out/r2_tests2.log | head -20
[ $rc -lt 124 ] || exit $rc
timeout -k 10 200 ./tools/ubench/tile_stream > gpurun_out/tile_stream.txt 2>&1; rc=$?; echo "tile_stream rc=$rc"; cat gpurun_out/tile_stream.txt
[ $rc -lt 124 ] || exit $rc
timeout -k 10 200 ./tools/ubench/store_patterns > gpurun_out/store_patterns.txt 2>&1; rc=$?; echo "store_patterns rc=$rc"; cat gpurun_out/store_patterns.txt
[ $rc -lt 124 ] || exit $rc
timeout -k 10 300 python tools/bench_layers.py 3 > gpurun_out/layers3.txt 2>&1; rc=$?; echo "layers rc=$rc"; cat gpurun_out/layers3.txt
[ $rc -lt 124 ] || exit $rc
SR_GEMM_NARROW=2 timeout -k 10 300 python tools/bench_layers.py 3 > gpurun_out/layers3_narrow.txt 2>&1; rc=$?; echo "layers narrow rc=$rc"; cat gpurun_out/layers3_narrow.txt
