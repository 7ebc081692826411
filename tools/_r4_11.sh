mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/full2.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4/full2.log
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench2_b6144.json 2> gpurun_out/r4/bench2_b6144.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/r4/bench2_b6144.json
timeout -k 10 200 python bench.py --global-batch 768 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-260
