set -e
python -m pytest tests/test_packed_roles_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > gpurun_out/r3_t2.log 2>&1 || { tail -40 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
for gb in 6144 768; do
  for pk in 0 1; do
    SR_PACK_ROLES=$pk python bench.py --global-batch $gb --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gb=$gb pack=$pk ms=%.2f loss=%s' % (d['ms_per_step'], d['config']['final_loss']))" >> gpurun_out/r3_pack.txt
  done
done
cat gpurun_out/r3_pack.txt
