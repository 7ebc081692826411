"""Which framework-side small GPU operations does one training step launch?  Runs the bench step under torch.profiler and lists
memcpy / fill / elementwise ops by count with the Python frames that issued them.  usage: python tools/small_ops.py [batch]"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from situation_recognition_amd import ops
from situation_recognition_amd.imsitu_encoder import imsitu_encoder
from situation_recognition_amd.model import FCGGNN
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dev = torch.device("cuda:0")
enc = imsitu_encoder.synthetic()
torch.manual_seed(1238)
net = FCGGNN(enc, 2048, steps=5, backbone=152, dtype=torch.bfloat16).to(dev)
net.train()
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adamax(params, lr=0.002)
img, verb, nouns = bench.synthetic_batch(enc, B, 224, dev)


def step():
    opt.zero_grad(set_to_none=True)
    pv, pn, pg = net(img, verb)
    loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
where = collections.defaultdict(collections.Counter)
for e in ev:
    n = e.name
    if n.startswith("aten::") and n in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy",
                                        "aten::zeros", "aten::add_", "aten::mul_", "aten::add", "aten::mul", "aten::cat", "aten::sum", "aten::empty_like"):
        cnt[n] += 1
        st = [f for f in (e.stack or []) if "situation_recognition_amd" in f or "bench" in f or "optim" in f or "clip_grad" in f]
        where[n][st[0] if st else "(other)"] += 1
for n, c in cnt.most_common():
    print("%-18s %5d" % (n, c))
    for f, k in where[n].most_common(8):
        print("      %4d  %s" % (k, f))
kern = collections.Counter(e.name for e in ev if e.device_type == torch.autograd.DeviceType.CUDA)
print("--- device activities by count")
for n, c in kern.most_common(25):
    print("%6d  %s" % (c, n[:110]))
