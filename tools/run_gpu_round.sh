set -o pipefail
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 -k "stem" 2>&1 | tail -2
timeout -k 10 200 python tools/stem_bench.py 2>&1 | grep -v amdgpu
timeout -k 10 200 python tools/bench_layers.py stem 2>&1 | grep -v amdgpu
