"""Batch statistics of the bottleneck expansion convs: Gram route (sr_gram + sr_bn_finalize_gram) against the
statistics-only conv launch + sr_bn_finalize, at the ResNet-152 layer shapes of a batch."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, hw, C in (("layer1", 56, 64), ("layer2", 28, 128), ("layer3", 14, 256), ("layer4", 7, 512)):
    M, N = B * hw * hw, 4 * C
    x = torch.randn(M, C, device="cuda").relu_().to(torch.bfloat16).view(B, hw, hw, C)
    w = (torch.randn(N, C, device="cuda") / C ** 0.5).to(torch.bfloat16)
    gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    part = ops.gram(x.view(-1, C))
    t_g = timed(lambda: ops.gram(x.view(-1, C)))
    t_f = timed(lambda: ops.bn_finalize_gram(part, w, M, gamma, beta, None, None, 0.1, 1e-5))
    st = ops.conv2d(x, w, N, 1, 1, 0, stats_only=True)
    t_c = timed(lambda: ops.conv2d(x, w, N, 1, 1, 0, stats_only=True))
    t_b = timed(lambda: ops.bn_finalize(st, M, gamma, beta, None, None, 0.1, 1e-5))
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    x2 = x.view(-1, C).clone()
    t_ap = timed(lambda: ops.bn_apply(x2, sc, sh, relu=True, out=x2))
    t_fu = timed(lambda: ops.bn_apply_gram(x2, sc, sh)) if C <= 256 else float("nan")
    print("   fused bn_apply+gram %7.1f us (%5.2f TB/s r+w)  vs  bn_apply %7.1f us + gram %7.1f us" % (t_fu, 4.0 * M * C / t_fu / 1e6, t_ap, t_g))
    s1, h1 = ops.bn_finalize_gram(part, w, M, gamma, beta, None, None, 0.1, 1e-5)
    s2, h2 = ops.bn_finalize(st, M, gamma, beta, None, None, 0.1, 1e-5)
    print("%s M=%9d C=%3d | gram %7.1f us (%5.2f TB/s of x) + finalize %6.1f us | stats-only conv %7.1f us + finalize %6.1f us | "
          "partials %d x %d | scale rel diff %.2e shift diff %.2e" %
          (name, M, C, t_g, 2.0 * M * C / t_g / 1e6, t_f, t_c, t_b, part.shape[0], part.shape[1],
           float(((s1 - s2).abs() / s2.abs()).max()), float((h1 - h2).abs().max())), flush=True)
