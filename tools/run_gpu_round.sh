set -o pipefail
mkdir -p gpurun_out
run() { tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r2_bench_$tag.json 2> gpurun_out/r2_bench_$tag.err; rc=$?
  echo "$tag rc=$rc $(python3 -c "
import json
d=json.load(open('gpurun_out/r2_bench_$tag.json')); print(d['value'], d['ms_per_step'])" 2>/dev/null)"
  [ $rc -lt 124 ] || exit $rc; }
run base SR_OVERLAP=0
run ov SR_OVERLAP=1
run ov_half SR_OVERLAP=1 SR_GEMM_HALF=1 SR_GEMM_NARROW=2
run half_noov SR_OVERLAP=0 SR_GEMM_HALF=1 SR_GEMM_NARROW=2
run ov_half_p2 SR_OVERLAP=1 SR_GEMM_HALF=1 SR_GEMM_NARROW=2 SR_PHASE_UNITS=2
run ov_half_p3 SR_OVERLAP=1 SR_GEMM_HALF=1 SR_GEMM_NARROW=2 SR_PHASE_UNITS=3
run ov_p3 SR_OVERLAP=1 SR_PHASE_UNITS=3
