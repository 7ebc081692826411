"""Experiment: does running a layer3 block's expansion conv and the NEXT block's reduce conv sub-batch by sub-batch keep the 1024-channel
block output in the memory-side cache (256 MB) for its second reader?  The reduce conv needs no statistics of its input (the block output
is final), so the pair may be interleaved over image ranges; its own partial statistics just come in more rows.
usage: python tools/pair_interleave.py [batch]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(5)
y2 = torch.randn(B, 14, 14, 256, device=dev, generator=g).to(dt)
zin = torch.randn(B, 14, 14, 1024, device=dev, generator=g).relu_().to(dt)
wexp = (torch.randn(1024, 256, device=dev, generator=g) * 0.06).to(dt)
wred = (torch.randn(256, 1024, device=dev, generator=g) * 0.03).to(dt)
sc2, sh2 = 0.5 + torch.rand(256, device=dev), 0.1 * torch.randn(256, device=dev)
sc3, sh3 = 0.5 + torch.rand(1024, device=dev), 0.1 * torch.randn(1024, device=dev)
zout = torch.empty_like(zin)
y1 = torch.empty(B, 14, 14, 256, device=dev, dtype=dt)


def pair(S):
    n = B // S
    for s in range(S):
        sl = slice(s * n, (s + 1) * n)
        ops.conv2d(y2[sl], wexp, 1024, 1, 1, 0, bias=sh3, escale=sc3, res=zin[sl], relu=True, in_affine=(sc2, sh2), out=zout[sl])
        ops.conv2d(zout[sl], wred, 256, 1, 1, 0, want_stats=True, out=y1[sl])


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pair(1)
ref = y1.clone()
for S in (1, 4, 1, 4, 3, 4, 1, 2, 6, 4, 1):
    if B % S:
        continue
    t = timed(lambda: pair(S))
    print("sub-batches %2d (%5d images, block output %6.1f MB each): pair %.1f us  bit-identical y1: %s" % (S, B // S, B // S * 196 * 2048 / 1e6, t, bool(torch.equal(y1, ref))), flush=True)
