mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/full3.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4/full3.log
timeout -k 10 300 python tools/layer_breakdown.py 6144 > gpurun_out/r4/layer_breakdown_b6144.txt 2>&1; head -22 gpurun_out/r4/layer_breakdown_b6144.txt | cut -c1-150
