set -e
for gb in 768 1536 3072 6144; do
  for sh in 1 2; do
    SR_BACKBONE_CU_SHARE=$sh python bench.py --global-batch $gb --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gb=$gb share=$sh ms=%.2f' % d['ms_per_step'])" >> gpurun_out/r3_share.txt
  done
done
cat gpurun_out/r3_share.txt
