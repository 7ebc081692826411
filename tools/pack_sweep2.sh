for gb in 1536 3072; do
  for pk in 0 1; do
    SR_PACK_ROLES=$pk python bench.py --global-batch $gb --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gb=$gb pack=$pk ms=%.2f' % d['ms_per_step'])"
  done
done
