set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/expand_dbg.txt
for d in 0 1 2 3 4 6 8 9 14 15; do
  SR_EXPAND_DEBUG=$d timeout -k 10 120 python tools/bench_expand.py 3 2>&1 | grep layer >> gpurun_out/expand_dbg.txt; rc=$?
  [ $rc -lt 124 ] || exit $rc
done
cat gpurun_out/expand_dbg.txt
