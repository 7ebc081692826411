"""Times train-mode (raw output + statistics) convolution launches of the backbone's hot shapes at a given batch.
usage: python tools/conv_time.py [batch]   (environment switches such as SR_GEMM_DEBUG are read once per process)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
for name, Cin, Cout, k, H in (("1x1 1024->256 @14", 1024, 256, 1, 14), ("3x3 256->256 @14", 256, 256, 3, 14), ("1x1 512->128 @28", 512, 128, 1, 28),
                              ("3x3 512->512 @7", 512, 512, 3, 7)):
    g = torch.Generator(device="cuda").manual_seed(Cin + k)
    x = torch.randn(B, H, H, Cin, device="cuda", generator=g).relu_().to(torch.bfloat16)
    w = (torch.randn(Cout, k * k * Cin, device="cuda", generator=g) * (k * k * Cin) ** -0.5).to(torch.bfloat16)
    us = timed(lambda: ops.conv2d(x, w, Cout, k, 1, k // 2, want_stats=True))
    out.append("%s %.1f us" % (name, us))
    del x, w
print("SR_GEMM_DEBUG=%s batch %d: " % (os.environ.get("SR_GEMM_DEBUG", "-"), B) + " | ".join(out), flush=True)
