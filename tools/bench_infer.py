#!/usr/bin/env python3
"""Single-image inference latency (the `results` path of the reference, sr.py:235-281): predict_verb + predict_nouns at batch 1,
eval mode, ResNet-152 + GGNN T=4, bf16; eager launches vs hipGraph replay of the backbone passes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd.imsitu_encoder import imsitu_encoder
from situation_recognition_amd.model import FCGGNN

torch.manual_seed(0)
net = FCGGNN(imsitu_encoder.synthetic(), 2048, steps=4, backbone=152, dtype=torch.bfloat16).cuda().eval()
img = torch.randn(1, 3, 224, 224).clamp_(-2.2, 2.7).cuda()

def once():
    with torch.no_grad():
        lv = net.predict_verb(img, 1)
        return lv, net.predict_nouns(img, torch.argmax(lv, 1), 1)

def lat(n=30):
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

a = once()
print("eager : %.2f ms per image" % lat())
net.enable_graphs()
b = once()
print("graphs: %.2f ms per image" % lat())
for x, y in zip(a, b):
    assert torch.equal(x, y), "graph replay differs from eager"
print("graph replay bit-identical to eager")
