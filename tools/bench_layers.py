#!/usr/bin/env python3
"""Times the exact launches of one ResNet-152 bottleneck per stage at the benchmark's batch (GPU box): the three
convolutions in their train-mode forms (reduce + statistics, 3x3 + statistics, expansion with scale/shift + residual + ReLU),
the BatchNorm sweeps between them, and the stem.  Prints us, TFLOP/s and algorithmic GB/s per launch.
    B=6144 python tools/bench_layers.py [stage ...]        stages: 1 2 3 4 stem (default: 3)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops

B = int(os.environ.get("B", "6144"))
dt, dev = torch.bfloat16, "cuda"
REP = int(os.environ.get("REP", "5"))


def timeit(fn, n=REP):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def line(name, us, flops, nbytes):
    print("%-44s %9.1f us  %7.1f TF/s  %7.0f GB/s" % (name, us, flops / us / 1e6, nbytes / us / 1e3), flush=True)


def stage(s):
    H = {1: 56, 2: 28, 3: 14, 4: 7}[s]
    C = 64 << (s - 1)
    M = B * H * H
    x4 = torch.randn(B, H, H, 4 * C, device=dev).relu_().to(dt)
    w1 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).to(dt)
    w2 = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).to(dt)
    w3 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).to(dt)
    sc4, sh4 = 0.5 + torch.rand(4 * C, device=dev), 0.1 * torch.randn(4 * C, device=dev)
    sc, sh = 0.5 + torch.rand(C, device=dev), 0.1 * torch.randn(C, device=dev)
    y1, _ = ops.conv2d(x4, w1, C, 1, 1, 0, want_stats=True)
    tag = "layer%d @%d C=%d" % (s, H, C)
    line(tag + " 1x1 %d->%d +stats" % (4 * C, C), timeit(lambda: ops.conv2d(x4, w1, C, 1, 1, 0, want_stats=True)), 2.0 * M * 4 * C * C, 2.0 * M * 5 * C)
    line(tag + " bn_apply", timeit(lambda: ops.bn_apply(y1, sc, sh, relu=True, out=y1)), 0, 4.0 * M * C)
    line(tag + " 3x3 %d->%d +stats" % (C, C), timeit(lambda: ops.conv2d(y1, w2, C, 3, 1, 1, want_stats=True)), 2.0 * M * 9 * C * C, 4.0 * M * C)
    if C <= 256:
        line(tag + " bn_apply+gram", timeit(lambda: ops.bn_apply_gram(y1.view(M, C), sc, sh)), 0, 4.0 * M * C)
    line(tag + " gram", timeit(lambda: ops.gram(y1.view(M, C))), 0, 2.0 * M * C)
    line(tag + " 1x1 %d->%d scale+res+relu" % (C, 4 * C), timeit(lambda: ops.conv2d(y1, w3, 4 * C, 1, 1, 0, bias=sh4, escale=sc4, res=x4, relu=True)),
         2.0 * M * 4 * C * C, 2.0 * M * 9 * C)
    line(tag + " 1x1 %d->%d plain store" % (C, 4 * C), timeit(lambda: ops.conv2d(y1, w3, 4 * C, 1, 1, 0)), 2.0 * M * 4 * C * C, 2.0 * M * 5 * C)
    line(tag + " 1x1 %d->%d stats only" % (C, 4 * C), timeit(lambda: ops.conv2d(y1, w3, 4 * C, 1, 1, 0, stats_only=True)), 2.0 * M * 4 * C * C, 2.0 * M * C)


def stem():
    img = torch.randn(B, 3, 224, 224, device=dev)
    w = (torch.randn(64, 256, device=dev) * 0.1).to(dt)
    sc, sh = 0.5 + torch.rand(64, device=dev), 0.1 * torch.randn(64, device=dev)
    line("stem_prep", timeit(lambda: ops.stem_prep(img, dt)), 0, B * 3 * 224 * 224 * 4 + B * 230 * 230 * 8)
    xp = ops.stem_prep(img, dt)
    M = B * 112 * 112
    line("stem 7x7/2 +stats", timeit(lambda: ops.conv2d(xp, w, 64, 7, 2, 3, want_stats=True, stem_hw=(224, 224))), 2.0 * M * 64 * 147, xp.numel() * 2 + M * 128)
    y, _ = ops.conv2d(xp, w, 64, 7, 2, 3, want_stats=True, stem_hw=(224, 224))
    line("maxpool (BN+ReLU fused)", timeit(lambda: ops.maxpool3x3s2(y, sc, sh)), 0, M * 128 + M // 4 * 128)


if __name__ == "__main__":
    print("B=%d SR_GEMM_NARROW=%s" % (B, os.environ.get("SR_GEMM_NARROW")))
    for a in (sys.argv[1:] or ["3"]):
        stem() if a == "stem" else stage(int(a))
