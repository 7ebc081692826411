#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM / implicit-conv kernel on the ResNet-152 and GGNN shapes (GPU box).
SR_GEMM_DEBUG=1 (no MFMA) / 2 (no loads) give the load-only / compute-only times of the same launch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops

B = int(os.environ.get("B", "768"))
dt = torch.bfloat16
dev = "cuda"

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

def conv_case(name, H, Cin, Cout, k, s, stats=True):
    x = torch.randn(B, H, H, Cin, device=dev).to(dt)
    w = (torch.randn(Cout, k * k * Cin, device=dev) * (k * k * Cin) ** -0.5).to(dt)
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    us = timeit(lambda: ops.conv2d(x, w, Cout, k, s, p, want_stats=stats))
    byts = (x.numel() + B * Ho * Ho * Cout) * 2
    print("%-34s M=%8d N=%5d K=%5d  %8.1f us  %7.1f TF/s  (min-HBM %6.1f us @5TB/s)" % (name, B*Ho*Ho, Cout, k*k*Cin, us, fl / us / 1e6, byts / 5e6), flush=True)

def gemm_case(name, M, N, K, pairs=1):
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(pairs)]
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(pairs)]
    fl = 2.0 * M * N * K * pairs
    us = timeit(lambda: ops.gemm(list(zip(As, Ws))))
    print("%-34s M=%8d N=%5d K=%5d  %8.1f us  %7.1f TF/s" % (name, M, N, K * pairs, us, fl / us / 1e6), flush=True)

print("SR_GEMM_DEBUG=%s B=%d" % (os.environ.get("SR_GEMM_DEBUG", "0"), B))
conv_case("l1 1x1 64->64", 56, 64, 64, 1, 1)
conv_case("l1 3x3 64->64", 56, 64, 64, 3, 1)
conv_case("l1 1x1 64->256", 56, 64, 256, 1, 1)
conv_case("l1 1x1 256->64", 56, 256, 64, 1, 1)
conv_case("l2 3x3 128->128", 28, 128, 128, 3, 1)
conv_case("l2 1x1 128->512", 28, 128, 512, 1, 1)
conv_case("l2 1x1 512->128", 28, 512, 128, 1, 1)
BB = B
B = int(os.environ.get("B3", "6144"))
conv_case("l3 1x1 1024->256", 14, 1024, 256, 1, 1)
conv_case("l3 3x3 256->256", 14, 256, 256, 3, 1)
conv_case("l3 1x1 256->1024", 14, 256, 1024, 1, 1)
conv_case("l4 3x3 512->512", 7, 512, 512, 3, 1)
conv_case("l4 1x1 512->2048", 7, 512, 2048, 1, 1)
gemm_case("ggnn n (M=B*6)", B * 6, 2048, 2048)
gemm_case("ggnn dW (K=M)", 2048, 2048, B * 6)
gemm_case("classifier", B * 6, 2001, 2048)
gemm_case("ggnn z 2-pair", B * 6, 2048, 2048, 2)
gemm_case("square 4096", 4096, 4096, 4096)
gemm_case("square 8192", 8192, 8192, 8192)
