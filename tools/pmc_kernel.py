"""Per-kernel sums of SQ counters from a rocprofv3 --pmc run: prints counter / dispatch for kernels whose name matches.
usage: pmc_kernel.py <counter_collection.csv> <name substring>"""
import collections, csv, sys
path, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if pat in k:
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    n = len(cnt[k])
    print(k[:110], "dispatches", n)
    for name, v in sorted(c.items()):
        print("   %-28s %16.0f per dispatch" % (name, v / n))
