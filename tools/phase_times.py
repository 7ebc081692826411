"""Where does a training step's time go at a given per-GPU batch?  Phases timed with a device synchronisation between them (which
removes the overlap ACROSS phases, not inside them): the two backbones (two streams), the heads' forward (verb path, predicted-verb and
ground-truth-verb noun branches), losses + backward, clip + Adamax; and the un-synchronised step beside their sum.
usage: python tools/phase_times.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from situation_recognition_amd import ops
from situation_recognition_amd.imsitu_encoder import imsitu_encoder
from situation_recognition_amd.model import FCGGNN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dev = torch.device("cuda", 0)
enc = imsitu_encoder.synthetic()
torch.manual_seed(1238)
net = FCGGNN(enc, 2048, steps=5, backbone=152, dtype=torch.bfloat16).to(dev).train()
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adamax(params, lr=0.002)
img, verb, nouns = bench.synthetic_batch(enc, B, 224, dev, seed_shift=0)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def full_step():
    opt.zero_grad(set_to_none=True)
    pv, pn, pg = net(img, verb)
    (net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)).backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()


for _ in range(3):
    full_step()
t0 = sync()
for _ in range(5):
    full_step()
t_full = (sync() - t0) / 5

# the same step with the heads cut off from the backbones by a synchronisation: FCGGNN.forward's two-stream backbone phase alone
acc = {}
main = torch.cuda.current_stream()
for it in range(5):
    opt.zero_grad(set_to_none=True)
    t = sync()
    side = net._side_streams.get(img.device) or torch.cuda.Stream()
    net._side_streams[img.device] = side
    prepped = net.convnet_verbs.prepare_input(img)
    side.wait_stream(main)
    prev = ops.set_cu_share(net.backbone_cu_share)
    with torch.cuda.stream(side):
        feat = net.convnet_nouns(img, bn_updates=2, prepped=prepped)
    feat_v = net.convnet_verbs(img, prepped=prepped)
    ops.set_cu_share(prev)
    t1 = sync(); acc["backbones (2 streams, half the CUs each)"] = acc.get("backbones (2 streams, half the CUs each)", 0) + t1 - t
    pv = net._verb_from_features(feat_v, B)
    t2 = sync(); acc["verb path forward"] = acc.get("verb path forward", 0) + t2 - t1
    pn = net._nouns_from_features(feat, torch.argmax(pv, 1), B)
    t3 = sync(); acc["predicted-verb noun branch forward"] = acc.get("predicted-verb noun branch forward", 0) + t3 - t2
    pg = net._nouns_from_features(feat, verb, B)
    t4 = sync(); acc["ground-truth noun branch forward"] = acc.get("ground-truth noun branch forward", 0) + t4 - t3
    loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
    t5 = sync(); acc["losses"] = acc.get("losses", 0) + t5 - t4
    loss.backward()
    t6 = sync(); acc["backward"] = acc.get("backward", 0) + t6 - t5
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    t7 = sync(); acc["clip + Adamax"] = acc.get("clip + Adamax", 0) + t7 - t6
print("batch %d: un-synchronised step %.2f ms; phases (synchronised):" % (B, 1e3 * t_full))
tot = 0
for k, v in acc.items():
    print("   %-44s %7.2f ms" % (k, 1e3 * v / 5)); tot += v / 5
print("   %-44s %7.2f ms" % ("sum", 1e3 * tot))
# one backbone alone on the whole chip
t = sync()
for _ in range(3):
    net.convnet_verbs(img)
print("   one backbone alone, whole chip: %.2f ms" % (1e3 * (sync() - t) / 3))
