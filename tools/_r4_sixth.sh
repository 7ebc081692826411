mkdir -p gpurun_out/r4
timeout -k 10 400 python -m pytest tests/test_model_gpu.py tests/test_packed_roles_gpu.py tests/test_parallel_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -5
for b in 768 6144; do timeout -k 10 200 python tools/phase_times.py $b 2>&1 | grep -v Warning | tail -12; done | tee gpurun_out/r4/phase_times_stackedtn.txt
timeout -k 10 200 python bench.py --global-batch 768 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-260
