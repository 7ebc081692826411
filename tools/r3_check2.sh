set -e
python -m pytest tests/test_model_gpu.py tests/test_packed_roles_gpu.py tests/test_parallel_gpu.py tests/test_scorer_results_gpu.py tests/test_driver_gpu.py -q -m gpu -x > gpurun_out/r3_t4.log 2>&1 || { tail -40 gpurun_out/r3_t4.log; exit 1; }
tail -2 gpurun_out/r3_t4.log
for gb in 768 1536 6144; do
  for og in 0 1; do
    SR_OVERLAP_GT=$og python bench.py --global-batch $gb --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gb=$gb overlap_gt=$og ms=%.2f' % d['ms_per_step'])"
  done
done
