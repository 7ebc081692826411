"""Launches the layer3 bottleneck's three convolutions alone (train-mode forms, benchmark batch) so that a `rocprofv3 --pmc ...`
run of this script attributes cache counters to one shape per kernel name + grid.  Which: argv[1] in {c3, reduce, expand, all}.
    cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d out -o x --output-format csv -- python3 $R/tools/pmc_layer3.py c3
Prints the hip-event time of each launch as well."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(os.environ.get("B", "6144"))
H, C = 14, 256
dt, dev = torch.bfloat16, "cuda"
x4 = torch.randn(B, H, H, 4 * C, device=dev).relu_().to(dt)
x1 = torch.randn(B, H, H, C, device=dev).relu_().to(dt)
w1 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).to(dt)
w2 = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).to(dt)
w3 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).to(dt)
sc4, sh4 = 0.5 + torch.rand(4 * C, device=dev), 0.1 * torch.randn(4 * C, device=dev)
runs = {"reduce": lambda: ops.conv2d(x4, w1, C, 1, 1, 0, want_stats=True),
        "c3": lambda: ops.conv2d(x1, w2, C, 3, 1, 1, want_stats=True),
        "expand": lambda: ops.conv2d(x1, w3, 4 * C, 1, 1, 0, bias=sh4, escale=sc4, res=x4, relu=True)}
for name, fn in runs.items():
    if which not in ("all", name):
        continue
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-8s %8.1f us" % (name, e0.elapsed_time(e1) / 3 * 1e3), flush=True)
