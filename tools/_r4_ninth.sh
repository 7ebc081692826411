mkdir -p gpurun_out/r4
timeout -k 10 240 python -m pytest tests/test_production_shapes_gpu.py -x -q -k "256_256_direct or direct_256" 2>&1 | tail -25
