timeout -k 10 300 python -m pytest tests/test_production_shapes_gpu.py -x -q -k "256_256_direct or direct_256 or generic_statistics" 2>&1 | tail -3
for v in 0 0; do SR_NO_C3_256=$v timeout -k 10 120 python tools/c3d256_time.py 6144 2>&1 | tail -1; done
