#!/usr/bin/env python3
"""Is the training step host-bound?  Time until the Python side has ENQUEUED a step (no synchronisation) against the time
until the GPU has finished it, per batch size."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import synthetic_batch
from situation_recognition_amd.imsitu_encoder import imsitu_encoder
from situation_recognition_amd.model import FCGGNN
dev = torch.device("cuda")
enc = imsitu_encoder.synthetic()
torch.manual_seed(1238)
net = FCGGNN(enc, 2048, steps=5, backbone=152, dtype=torch.bfloat16).to(dev).train()
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adamax(params, lr=0.002)
for B in [int(a) for a in (sys.argv[1:] or ["768", "6144"])]:
    img, verb, nouns = synthetic_batch(enc, B, 224, dev)
    def step():
        opt.zero_grad(set_to_none=True)
        pv, pn, pg = net(img, verb)
        loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    te, tt = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        te.append(t1 - t0); tt.append(t2 - t0)
    print("B=%d: enqueue %.1f ms, step %.1f ms" % (B, 1e3 * min(te), 1e3 * min(tt)), flush=True)
