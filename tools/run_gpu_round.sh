set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 -k "stem or pools or two_phase" > gpurun_out/r2_tests_stem.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/r2_tests_stem.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/bench_layers.py stem 2>&1 | grep -v amdgpu
SR_NO_STEM_DIRECT=1 timeout -k 10 200 python tools/bench_layers.py stem 2>&1 | grep "stem 7x7"
