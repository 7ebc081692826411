"""Round-3 counter targets, each launched alone a few times at the benchmark's layer3 shape (batch 6144, 14 x 14):
  reduce   1x1 reduce conv 1024 -> 256 with statistics  (conv_igemm_v3_kernel<bf16,bf16,4,1>)
  bngram   BatchNorm + ReLU + Gram sweep of the 256-channel tensor, no write-back (gram_kernel<256,false,false,true,false>)
  gram     plain Gram sweep (gram_kernel<256,false,false,false,true>)
usage: pmc_r3.py {reduce|bngram|gram|all}   (under `rocprofv3 --pmc ...`, one SQ pass per run; prints HIP-event times)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "all"
B, H, C = int(os.environ.get("B", "6144")), 14, 256
dt, dev = torch.bfloat16, "cuda"
M = B * H * H
runs = {}
if which in ("all", "reduce"):
    x4 = torch.randn(B, H, H, 4 * C, device=dev).relu_().to(dt)
    w1 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).to(dt)
    runs["reduce"] = (lambda: ops.conv2d(x4, w1, C, 1, 1, 0, want_stats=True), 2.0 * M * C * 4 * C, 2.0 * (M * 5 * C))
if which in ("all", "bngram", "gram"):
    y2 = torch.randn(M, C, device=dev).to(dt)
    sc, sh = 0.5 + torch.rand(C, device=dev), 0.1 * torch.randn(C, device=dev)
    if which != "gram":
        runs["bngram"] = (lambda: ops.bn_gram(y2, sc, sh), M * C * (C + 16.0), 2.0 * M * C)
    if which != "bngram":
        runs["gram"] = (lambda: ops.gram(y2), M * C * (C + 16.0), 2.0 * M * C)
for name, (fn, flops, nbytes) in runs.items():
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print("%-8s %8.1f us  %7.1f TFLOP/s  %6.2f TB/s algorithmic" % (name, us, flops / us / 1e6, nbytes / us / 1e6), flush=True)
