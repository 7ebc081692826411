"""A few launches of the BatchNorm + Gram sweep of layer3 (bn_gram: 256 channels, M = batch x 196, no write-back) for counter passes
(tools/pmc_sq.sh).  usage: python tools/pmc_gram.py [batch]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
x = torch.randn(B * 196, 256, device="cuda").to(torch.bfloat16)
sc, sh = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.1
for _ in range(3):
    ops.bn_gram(x, sc, sh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.bn_gram(x, sc, sh)
e1.record(); torch.cuda.synchronize()
print("bn_gram C=256 M=%d: %.1f us per sweep (%.2f TB/s)" % (B * 196, e0.elapsed_time(e1) * 100, B * 196 * 512 / (e0.elapsed_time(e1) * 100) / 1e6))
