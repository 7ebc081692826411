set -o pipefail
mkdir -p gpurun_out/r4
R=$(pwd)
for d in 0 128 0 128; do SR_GEMM_DEBUG=$d timeout -k 10 120 python tools/conv_time.py 6144 2>&1 | tail -1; done | tee gpurun_out/r4/rot_ab.txt
SR_GEMM_DEBUG=128 timeout -k 10 300 python -m pytest tests/test_production_shapes_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_full_configs_gpu.py -x -q -s -k config3 > gpurun_out/r4/config3.log 2>&1; grep -h "GATED\|REPORTED\|passed\|failed\|moves the pooled\|Error" gpurun_out/r4/config3.log
timeout -k 10 300 python -m pytest tests/test_production_shapes_gpu.py tests/test_kernels_gpu.py tests/test_race_screen_gpu.py tests/test_parallel_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
for sh in c3 reduce; do
timeout -k 10 300 bash tools/pmc_sq.sh r4_$sh "conv_igemm_v3" tools/pmc_layer3.py $sh > gpurun_out/r4/pmc_$sh.log 2>&1; echo "pmc $sh rc=$?"
done
