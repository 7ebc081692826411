set -o pipefail
mkdir -p gpurun_out/r4
(rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > gpurun_out/r4/sq_counters.txt) || true
hipcc --offload-arch=gfx950 -O2 tools/ubench/buffer_lds_oob.hip -o /tmp/buffer_lds_oob && /tmp/buffer_lds_oob > gpurun_out/r4/buffer_lds_oob.txt 2>&1; echo "oob rc=$?"; cat gpurun_out/r4/buffer_lds_oob.txt
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r4/full0.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/full0.log
timeout -k 10 400 bash tools/pmc_sq.sh r4_l3 "conv_igemm_v3|conv1x1_ws" tools/pmc_layer3.py all > gpurun_out/r4/pmc_l3.log 2>&1; echo "pmc rc=$?"; tail -60 gpurun_out/r4/pmc_l3.log
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench0_b6144.json 2> gpurun_out/r4/bench0_b6144.err; echo "bench rc=$?"; cat gpurun_out/r4/bench0_b6144.json | cut -c1-400
timeout -k 10 200 python bench.py --global-batch 768 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r4/bench0_b768.json 2>/dev/null; cat gpurun_out/r4/bench0_b768.json | cut -c1-300
