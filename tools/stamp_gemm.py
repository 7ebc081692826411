#!/usr/bin/env python3
"""In-kernel time shares of the v3 GEMM loop (run with SR_GEMM_DEBUG=4 on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from situation_recognition_amd import ops, _lib
dt, dev = torch.bfloat16, "cuda"

def report(name):
    buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
    _lib.check(_lib.lib().sr_debug_stamps(buf, 256 * 8 * 8), "stamps")
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    steps = a[:, :, 5].mean()
    seg = a[:, :, :5].mean(axis=(0, 1)) / max(steps, 1)
    tot = seg.sum()
    print("%-28s steps/wg %6.0f  cycles/step: wait %6.0f  barrier %6.0f  issue %6.0f  mfma %6.0f  epilogue %6.0f  total %6.0f" %
          (name, steps, *seg, tot), flush=True)
    w = a[:, :, :5].mean(axis=0) / max(steps, 1)
    print("   per wave mfma:", np.round(w[:, 3]).tolist(), " barrier:", np.round(w[:, 1]).tolist(), " wait:", np.round(w[:, 0]).tolist())

def gemm(M, N, K):
    A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    ops.gemm([(A, W)]); torch.cuda.synchronize(); ops.gemm([(A, W)])
    report("gemm %dx%dx%d" % (M, N, K))

def conv(B, H, Cin, Cout, k):
    x = torch.randn(B, H, H, Cin, device=dev).to(dt); w = (torch.randn(Cout, k * k * Cin, device=dev) * 0.02).to(dt)
    ops.conv2d(x, w, Cout, k, 1, k // 2, want_stats=True); torch.cuda.synchronize(); ops.conv2d(x, w, Cout, k, 1, k // 2, want_stats=True)
    report("conv %dx%d %d->%d k%d" % (H, H, Cin, Cout, k))

gemm(8192, 8192, 8192)
gemm(4096, 4096, 4096)
conv(768, 14, 256, 256, 3)
conv(768, 14, 1024, 256, 1)
conv(768, 14, 256, 1024, 1)
