"""Times the 128-channel direct 3x3 (layer2 shape, batch 6144) in its five forms."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
B = 6144
x = torch.randn(B, 28, 28, 128, device='cuda').relu_().to(torch.bfloat16)
w = (torch.randn(128, 9 * 128, device='cuda') * 0.03).to(torch.bfloat16)
sc, sh = 0.5 + torch.rand(128, device='cuda'), 0.1 * torch.randn(128, device='cuda')
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * 784 * 128 * 1152
for name, fn in (("direct stats", lambda: ops.conv2d(x, w, 128, 3, 1, 1, want_stats=True)),
                 ("direct stats + BN on load", lambda: ops.conv2d(x, w, 128, 3, 1, 1, want_stats=True, in_affine=(sc, sh))),
                 ("direct eval bias+relu", lambda: ops.conv2d(x, w, 128, 3, 1, 1, bias=sh, relu=True)),
                 ("direct plain (no stats)", lambda: ops.conv2d(x, w, 128, 3, 1, 1)),
                 ("direct stats only (no store)", lambda: ops.conv2d(x, w, 128, 3, 1, 1, stats_only=True))):
    us = timed(fn); print("%-28s %8.1f us %7.1f TFLOP/s" % (name, us, fl / us / 1e6), flush=True)
