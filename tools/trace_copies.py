"""Which Python call sites issue device-to-device copies / fills in a training step? (torch.profiler, with stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from situation_recognition_amd.imsitu_encoder import imsitu_encoder
from situation_recognition_amd.model import FCGGNN
B = 96
enc = imsitu_encoder.synthetic()
net = FCGGNN(enc, 2048, steps=5, backbone=152, dtype=torch.bfloat16).cuda().train()
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adamax(params, lr=0.002)
img = torch.randn(B, 3, 224, 224, device="cuda"); verb = torch.randint(0, 504, (B,), device="cuda")
nouns = torch.randint(0, 2001, (B, 3, 6), device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    pv, pn, pg = net(img, verb)
    loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "situation_recognition_amd" in s or "bench" in s or "torch/optim" in s or "clip_grad" in s]
        cnt[(e.name, st[0] if st else "(other)")] += 1
for (n, s), c in cnt.most_common(25):
    print("%5d  %-18s %s" % (c, n, s[-110:]))
