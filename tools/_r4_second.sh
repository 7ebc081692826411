set -o pipefail
mkdir -p gpurun_out/r4
R=$(pwd)
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4/full1.log 2>&1; echo "tests rc=$?"; tail -40 gpurun_out/r4/full1.log | grep -v Warning
grep -h "GATED\|REPORTED" gpurun_out/r4/full1.log
timeout -k 10 300 python -m pytest tests/test_full_configs_gpu.py -x -q -s -k config3 > gpurun_out/r4/config3.log 2>&1; grep -h "GATED\|REPORTED\|passed\|failed" gpurun_out/r4/config3.log
timeout -k 10 400 bash tools/pmc_sq.sh r4_l3 "conv_igemm_v3|conv1x1_ws" tools/pmc_layer3.py all > gpurun_out/r4/pmc_l3.log 2>&1; echo "pmc rc=$?"; cat gpurun_out/pmc_r4_l3/summary.txt
for pm in 1 0; do
  SR_WS_PLAIN_MAP=$pm python3 tools/pmc_layer3.py expand
  (cd /tmp && export TMPDIR=/tmp && SR_WS_PLAIN_MAP=$pm rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r4/fetch_pm$pm -o p --output-format csv -- python3 $R/tools/pmc_layer3.py expand > /dev/null 2>&1)
  python3 tools/pmc_kernel.py $(find gpurun_out/r4/fetch_pm$pm -name "*counter_collection.csv" | head -1) conv1x1_ws
done
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench1_b6144.json 2> gpurun_out/r4/bench1_b6144.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/r4/bench1_b6144.json
