#!/usr/bin/env python3
"""Times the layer3 / layer2 expansion convolutions (scale + residual + ReLU) only; SR_EXPAND_DEBUG selects diagnostic variants."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
from bench_layers import timeit, line, B, dt, dev
for s in [int(a) for a in (sys.argv[1:] or ["3"])]:
    H = {1: 56, 2: 28, 3: 14, 4: 7}[s]; C = 64 << (s - 1); M = B * H * H
    x4 = torch.randn(B, H, H, 4 * C, device=dev).relu_().to(dt)
    y1 = torch.randn(B, H, H, C, device=dev).relu_().to(dt)
    w3 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).to(dt)
    sc4, sh4 = 0.5 + torch.rand(4 * C, device=dev), 0.1 * torch.randn(4 * C, device=dev)
    tag = "dbg=%s layer%d" % (os.environ.get("SR_EXPAND_DEBUG", "0"), s)
    line(tag + " 1x1 %d->%d scale+res+relu" % (C, 4 * C), timeit(lambda: ops.conv2d(y1, w3, 4 * C, 1, 1, 0, bias=sh4, escale=sc4, res=x4, relu=True)), 2.0 * M * 4 * C * C, 2.0 * M * 9 * C)
    line(tag + " 1x1 %d->%d plain" % (C, 4 * C), timeit(lambda: ops.conv2d(y1, w3, 4 * C, 1, 1, 0)), 2.0 * M * 4 * C * C, 2.0 * M * 5 * C)
