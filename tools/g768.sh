for cfg in "off 0" "on 0" "off 1" "on 1"; do
  set -- $cfg
  SR_PACK_ROLES=$2 python bench.py --global-batch 768 --graphs $1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graphs=$1 pack=$2 ms=%.2f' % d['ms_per_step'])"
done
