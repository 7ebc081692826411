for b in 768 6144; do timeout -k 10 200 python tools/phase_times.py $b 2>&1 | grep -v Warning | tail -14; done | tee gpurun_out/r4/phase_times.txt
