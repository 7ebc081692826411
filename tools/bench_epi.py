import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 6144
for (H, Cin, Cout) in ((14, 256, 1024), (14, 1024, 256), (56, 64, 256)):
    Bq = B if H == 14 else 768
    x = torch.randn(Bq, H, H, Cin, device=dev).to(dt); w = (torch.randn(Cout, Cin, device=dev) * 0.05).to(dt)
    y = torch.empty(Bq, H, H, Cout, device=dev, dtype=dt); idn = torch.randn(Bq, H, H, Cout, device=dev).to(dt)
    sc, sh = torch.rand(Cout, device=dev), torch.rand(Cout, device=dev)
    M = Bq * H * H
    print("shape M=%d K=%d N=%d  out %.0f MB in %.0f MB" % (M, Cin, Cout, M * Cout * 2 / 1e6, M * Cin * 2 / 1e6))
    print("  conv + stats      %8.1f us" % timeit(lambda: ops.conv2d(x, w, Cout, 1, 1, 0, want_stats=True, out=y)))
    print("  conv no stats     %8.1f us" % timeit(lambda: ops.conv2d(x, w, Cout, 1, 1, 0, out=y)))
    print("  stats only        %8.1f us" % timeit(lambda: ops.conv2d(x, w, Cout, 1, 1, 0, stats_only=True)))
    print("  scale+shift+relu  %8.1f us" % timeit(lambda: ops.conv2d(x, w, Cout, 1, 1, 0, bias=sh, escale=sc, relu=True, out=y)))
    print("  ... + residual    %8.1f us" % timeit(lambda: ops.conv2d(x, w, Cout, 1, 1, 0, bias=sh, escale=sc, res=idn, relu=True, out=y)))
    print("  bn_apply(out)     %8.1f us" % timeit(lambda: ops.bn_apply(y, sc, sh, relu=True, out=y)))
    print("  bn_apply(out,res) %8.1f us" % timeit(lambda: ops.bn_apply(y, sc, sh, res=idn, relu=True, out=y)))
    print("  torch copy out    %8.1f us" % timeit(lambda: y.copy_(idn)))
