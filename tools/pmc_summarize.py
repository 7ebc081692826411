"""Summarises rocprofv3 outputs of `bench.py` into profiles/: per-kernel time table from the kernel trace, and the conv
kernels' HBM bytes per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x2 on gfx950 for wide
coalesced reads, WRITE_SIZE exact: MI355X_MICROARCH.md, HBM section).
usage: pmc_summarize.py <kernel_trace.csv> <fetch_counter.csv> <write_counter.csv> <out_dir> <tag>"""
import collections, csv, json, os, sys

trace, fetch, write, out, tag = sys.argv[1:6]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(trace)):
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
with open(os.path.join(out, "bench_%s_kernel_stats.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, n, t, "%.1f" % (t / n), "%.2f" % (100.0 * t / tot)])
CONV = ("conv_igemm", "conv1x1_expand", "conv1x1_ws", "conv1x1_pair", "stem_conv_kernel", "stem_pool_kernel", "conv3x3_fp8", "conv3x3_c64", "conv3x3_c128", "conv3x3_c256", "conv3x3_slices")   # the backbone convolution kernels
conv = {k: v for k, v in agg.items() if any(c in k for c in CONV)}
n_conv, t_conv = sum(v[0] for v in conv.values()), sum(v[1] for v in conv.values())


def counter_sum(path, name):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and any(c in r["Kernel_Name"] for c in CONV):
            s += float(r["Counter_Value"]); n += 1
    return s, n


def by_kernel(path, name):
    """per-kernel totals of one counter (all kernels of the run) -> pmc_<counter>_<tag>_by_kernel.csv"""
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            a = per[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    with open(os.path.join(out, "pmc_%s_%s_by_kernel.csv" % (name.lower(), tag)), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", name + "_KiB_sum", "KiB_per_dispatch"])
        for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, n, "%.1f" % v, "%.1f" % (v / n)])


by_kernel(fetch, "FETCH_SIZE")
by_kernel(write, "WRITE_SIZE")
fs, fn = counter_sum(fetch, "FETCH_SIZE")
ws, wn = counter_sum(write, "WRITE_SIZE")
res = {
    "kernel": "backbone convolutions: conv_igemm_* + conv1x1_pair_kernel + conv1x1_ws_kernel + conv3x3_c64_kernel + conv3x3_slices_kernel + stem_conv/stem_pool_kernel (all variants)",
    "launches_in_trace": n_conv, "avg_launch_us_in_trace": t_conv / n_conv / 1e3,
    "fetch_size_kib_sum": fs, "fetch_launches": fn, "write_size_kib_sum": ws, "write_launches": wn,
    "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE exact; separate --pmc passes",
    "hbm_bytes_per_launch": (2.0 * fs / fn + ws / wn) * 1024.0,
}
json.dump(res, open(os.path.join(out, "conv_traffic_%s.json" % tag), "w"), indent=1)
print(json.dumps(res, indent=1))
