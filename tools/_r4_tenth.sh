mkdir -p gpurun_out/r4
for v in 0 1 0 1; do SR_NO_C3_256=$v timeout -k 10 120 python tools/c3d256_time.py 6144 2>&1 | tail -1; done | tee gpurun_out/r4/c3d256_time.txt
SR_NO_C3_256=0 timeout -k 10 120 python tools/c3d256_time.py 768 2>&1 | tail -1
SR_NO_C3_256=1 timeout -k 10 120 python tools/c3d256_time.py 768 2>&1 | tail -1
