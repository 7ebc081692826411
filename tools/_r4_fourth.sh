set -o pipefail
mkdir -p gpurun_out/r4
R=$(pwd)
timeout -k 10 300 python -m pytest tests/test_production_shapes_gpu.py tests/test_kernels_gpu.py tests/test_race_screen_gpu.py tests/test_parallel_gpu.py tests/test_model_gpu.py tests/test_packed_roles_gpu.py -x -q 2>&1 | tail -3
timeout -k 10 200 python tools/trace_copies.py > gpurun_out/r4/trace_copies.txt 2>&1; tail -40 gpurun_out/r4/trace_copies.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r4/trace768 -o t --output-format csv -- python3 $R/bench.py --global-batch 768 --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/r4/trace768.json 2> $R/gpurun_out/r4/trace768.err); echo "trace rc=$?"
python3 - <<'PY'
import csv, glob, gzip
f = glob.glob("gpurun_out/r4/trace768/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
# keep a compact copy: start, end, stream/queue, short name, grid
with gzip.open("gpurun_out/r4/trace768_compact.csv.gz", "wt") as o:
    w = csv.writer(o)
    w.writerow(["start", "end", "queue", "stream", "name", "grid"])
    for r in rows:
        w.writerow([r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:70], r.get("Grid_Size", "")])
PY
rm -rf gpurun_out/r4/trace768
ls -la gpurun_out/r4/
