"""Times train-mode (raw output + statistics) convolution launches of the backbone's hot shapes at a given batch, and checks
each against an fp32 reference on its first images plus a checksum of the whole output (two runs under different environment
switches -- SR_GEMM_DEBUG, SR_GEMM_SPLIT: read once per process -- must print the same checksum).
usage: python tools/conv_time.py [batch]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
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
                              ("1x1 2048->512 @7", 2048, 512, 1, 7), ("1x1 1024->512 @14", 1024, 512, 1, 14), ("1x1 512->256 @28", 512, 256, 1, 28)):
    g = torch.Generator(device="cuda").manual_seed(Cin + k)
    x = torch.randn(B, H, H, Cin, device="cuda", generator=g).relu_().to(torch.bfloat16)
    w = (torch.randn(Cout, k * k * Cin, device="cuda", generator=g) * (k * k * Cin) ** -0.5).to(torch.bfloat16)
    y, st = ops.conv2d(x, w, Cout, k, 1, k // 2, want_stats=True)
    n = 4
    ref = F.conv2d(x[:n].float().permute(0, 3, 1, 2), w.float().view(Cout, k, k, Cin).permute(0, 3, 1, 2), padding=k // 2).permute(0, 2, 3, 1)
    err = float((y[:n].float() - ref).abs().max()) / float(ref.abs().max())
    ysum = int(y.view(torch.int16).to(torch.int64).sum())
    s1 = st.view(-1, 2, Cout)[:, 0].double().sum(0)
    want1 = y.float().view(-1, Cout).double().sum(0)
    serr = float((s1 - want1).abs().max() / want1.abs().max())
    assert err < 1.2e-2 and serr < 1e-3, (name, err, serr)
    us = timed(lambda: ops.conv2d(x, w, Cout, k, 1, k // 2, want_stats=True))
    out.append("%s %.1f us (ysum %d)" % (name, us, ysum % 1000003))
    del x, w, y, st
print("DEBUG=%s SPLIT=%s batch %d: " % (os.environ.get("SR_GEMM_DEBUG", "-"), os.environ.get("SR_GEMM_SPLIT", "-"), B) + " | ".join(out), flush=True)
