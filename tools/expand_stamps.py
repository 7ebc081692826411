#!/usr/bin/env python3
"""In-kernel section times of conv1x1_expand_kernel (diagnostic build: build.py --stamps, SR_LIB_PATH=.../libsrhip_stamps.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from situation_recognition_amd import ops
B = int(os.environ.get("B", "6144"))
H, C = 14, 256
M = B * H * H
dt, dev = torch.bfloat16, "cuda"
x4 = torch.randn(B, H, H, 4 * C, device=dev).relu_().to(dt)
y1 = torch.randn(B, H, H, C, device=dev).relu_().to(dt)
w3 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).to(dt)
sc4, sh4 = 0.5 + torch.rand(4 * C, device=dev), 0.1 * torch.randn(4 * C, device=dev)
for _ in range(3):
    ops.conv2d(y1, w3, 4 * C, 1, 1, 0, bias=sh4, escale=sc4, res=x4, relu=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
lib = ops.lib()
lib.srx_expand_stamps.argtypes = [ctypes.c_void_p]
assert lib.srx_expand_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
steps = a[:, :, 7].mean()
names = ["wait vmcnt + barrier", "LDS-DMA issue", "fragment reads + wait", "MFMAs", "staging write/read + wait", "epilogue arithmetic", "store + residual load issue"]
tot = a[:, :, :7].sum(2).mean()
print("K-steps per wave %.0f, stamped cycles per step %.0f" % (steps, tot / steps))
for k, n in enumerate(names):
    print("  %-30s %7.1f cycles/step  (waves 0-3: %7.1f, waves 4-7: %7.1f)" % (n, a[:, :, k].mean() / steps, a[:, :4, k].mean() / steps, a[:, 4:, k].mean() / steps))
