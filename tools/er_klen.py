"""Does the K loop of the 1x1 expansion conv (scale/shift + residual + ReLU epilogue) overlap with its store phase?
Same output tensor, K = 64 / 128 / 256 / 512: if the time grows by the K loop's length, it does not."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
dt, dev = torch.bfloat16, "cuda"
B, H, N = 6144, 14, 1024
res = torch.randn(B, H, H, N, device=dev).to(dt)
sc, sh = torch.rand(N, device=dev), torch.rand(N, device=dev)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K in (64, 128, 256, 512):
    x = torch.randn(B, H, H, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    t1 = timed(lambda: ops.conv2d(x, w, N, 1, 1, 0, bias=sh, escale=sc, res=res, relu=True))
    t2 = timed(lambda: ops.conv2d(x, w, N, 1, 1, 0, stats_only=True))
    print("K=%4d  fused epilogue %7.1f us   statistics-only (K loop + light epilogue) %7.1f us" % (K, t1, t2), flush=True)
