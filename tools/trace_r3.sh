# kernel traces (no counters) of the benchmark at two per-GPU batch sizes -> per-kernel totals
set -e
R=$(pwd); OUT=$R/gpurun_out/trace_r3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for gb in ${BATCHES:-6144 768}; do
  rocprofv3 --kernel-trace -d $OUT/b$gb -o t --output-format csv -- python3 $R/bench.py --global-batch $gb --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/b$gb.json 2> $OUT/b$gb.err || { tail -5 $OUT/b$gb.err; exit 1; }
  python3 - $OUT/b$gb $gb <<'PY'
import csv, collections, glob, sys
d, gb = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0])
rows = list(csv.DictReader(open(f)))
for r in rows:
    a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
with open(d + "_kernel_stats.csv", "w") as o:
    w = csv.writer(o); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, n, t, "%.1f" % (t / n), "%.2f" % (100.0 * t / tot)])
print("batch", gb, "kernels", len(rows), "sum of kernel time %.1f ms over 8 steps" % (tot / 1e6))
PY
  cat $OUT/b$gb.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
  find $OUT/b$gb -name "*.csv" -size +1M -delete
done
