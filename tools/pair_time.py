"""The fused expansion -> next-block-reduce launch (sr_conv_pair, csrc/pair.hip) of layer3 against the two launches it replaces
(weight-stationary expansion conv + generic reduce conv with statistics): bitwise comparison of both outputs, statistics against the
unfused launch's and an fp64 reference on a row sample, then timing of both forms (alternating).
usage: python tools/pair_time.py [batch] [check|time|both] [l3|l2|l1|l23|l12]     (l23 / l12: a layer's last block + the next layer's conv1, 2 C outputs)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
H, C, CR = {"l3": (14, 256, 256), "l2": (28, 128, 128), "l1": (56, 64, 64), "l23": (28, 128, 256), "l12": (56, 64, 128)}[sys.argv[3] if len(sys.argv) > 3 else "l3"]
CX = 4 * C
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(B, H, H, C, device=dev, generator=g).to(dt)                       # raw 3x3 output
res = torch.relu(torch.randn(B, H, H, CX, device=dev, generator=g)).to(dt)        # identity (a block output: post-ReLU)
w3 = (torch.randn(CX, C, device=dev, generator=g) * C ** -0.5).to(dt)
w1 = (torch.randn(CR, CX, device=dev, generator=g) * CX ** -0.5).to(dt)
insc, insh = 0.5 + torch.rand(C, device=dev), 0.1 * torch.randn(C, device=dev)
esc, esh = 0.2 + 0.3 * torch.rand(CX, device=dev), 0.1 * torch.randn(CX, device=dev)
M = B * H * H
wp = ops.conv_pair_pack(w3, w1)


def unfused():
    z = ops.conv2d(x, w3, CX, 1, 1, 0, bias=esh, escale=esc, res=res, relu=True, in_affine=(insc, insh))
    y, st = ops.conv2d(z, w1, CR, 1, 1, 0, want_stats=True)
    return z, y, st


def fused():
    return ops.conv_pair(x, wp, res, esc, esh, in_affine=(insc, insh))


if mode in ("check", "both"):
    z0, y0, st0 = unfused()
    z1, y1, st1 = fused()
    torch.cuda.synchronize()
    print("routes: expansion %s, reduce %s" % (ops.conv_route(B, H, H, C, CX, 1, 1, 0, res=True, relu=True, bias=True, escale=True, in_affine=True),
                                               ops.conv_route(B, H, H, CX, CR, 1, 1, 0, want_stats=True)))
    dz = (z0.view(torch.int16) != z1.view(torch.int16)).sum().item()
    dy = (y0.view(torch.int16) != y1.view(torch.int16)).sum().item()
    print("z: %d of %d elements differ from the unfused expansion conv; max |diff| %.4g" % (dz, z0.numel(), (z0.float() - z1.float()).abs().max().item()))
    print("y: %d of %d elements differ from the unfused reduce conv;    max |diff| %.4g" % (dy, y0.numel(), (y0.float() - y1.float()).abs().max().item()))
    s0, s1 = st0.double().sum(0), st1.double().sum(0)
    print("statistics vs unfused: sum rel %.3g, sumsq rel %.3g" % (((s0[0] - s1[0]).abs().max() / s0[0].abs().max()).item(),
                                                                    ((s0[1] - s1[1]).abs().max() / s0[1].abs().max()).item()))
    # fp32 reference on a row sample (first / last rows and a random set)
    idx = torch.cat([torch.arange(0, 512, device=dev), torch.arange(M - 512, M, device=dev), torch.randint(0, M, (3072,), device=dev, generator=g)])
    xa = torch.relu(x.view(M, C)[idx].float() * insc + insh).to(dt).float()
    zr = torch.relu(xa @ w3.float().t() * esc + esh + res.view(M, CX)[idx].float())
    print("z vs fp32 reference: max |diff| %.4g (range %.3g)" % ((zr - z1.view(M, CX)[idx].float()).abs().max().item(), zr.abs().max().item()))
    yr = z1.view(M, CX)[idx].float() @ w1.float().t()
    print("y vs fp32 reference (of the stored z): max |diff| %.4g (range %.3g)" % ((yr - y1.view(M, CR)[idx].float()).abs().max().item(), yr.abs().max().item()))
    yf = z1.view(M, CX).float() @ w1.float().t() if M * CX <= 2000000 * 1024 else None
    if yf is not None:
        print("statistics vs fp64 sums of the fp32 product: sum rel %.3g, sumsq rel %.3g" % (
            ((yf.double().sum(0) - s1[0]).abs().max() / s1[0].abs().max()).item(), (((yf.double() ** 2).sum(0) - s1[1]).abs().max() / s1[1].abs().max()).item()))
    del yf


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if mode in ("time", "both"):
    for rep in range(3):
        tu, tf = timed(unfused), timed(fused)
        gb = 2.0 * (M * C + M * CR + 2 * M * CX) / 1e9
        print("batch %d: unfused pair %.1f us | fused %.1f us (%.2f TB/s of its %.2f GB, %.0f TFLOP/s)" % (B, tu, tf, gb / tf * 1e3, gb, 2.0 * M * (C + CR) * CX / tf / 1e6), flush=True)
