"""Runs one GEMM / conv shape a few times (for rocprofv3 --pmc passes).  usage: one_gemm.py gemm M N K | conv3 B"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
dt, dev = torch.bfloat16, "cuda"
if sys.argv[1] == "gemm":
    M, N, K = map(int, sys.argv[2:5])
    a = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt)
    fn = lambda: ops.gemm([(a, w)])
else:
    B = int(sys.argv[2])
    x = torch.randn(B, 14, 14, 256, device=dev).to(dt); w = (torch.randn(256, 9 * 256, device=dev) * .05).to(dt)
    fn = lambda: ops.conv2d(x, w, 256, 3, 1, 1, want_stats=True)
for _ in range(3):
    fn()
torch.cuda.synchronize()
