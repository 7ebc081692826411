import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from golden_util import load, oracle_fcggnn, overfitting_json, sub
import situation_recognition_amd.model as m
from situation_recognition_amd.imsitu_encoder import imsitu_encoder
g3, g = load("g3_fcggnn_basic.npz"), load("g5_train_step.npz")
enc = imsitu_encoder(overfitting_json(), quiet=True)
net = m.FCGGNN(enc, int(g3["D"]), steps=4, backbone=int(g3["cfg_depth"]), dtype=torch.float32, width=int(g3["cfg_width"]), blocks=tuple(int(b) for b in g3["cfg_blocks"]))
net.load_state_dict(sub(g3, "state/"), strict=True); net.cuda(); net.train()
net.verb_classifier[0].p = 0.0; net.nouns_classifier[0].p = 0.0
img, verb, nouns = (torch.from_numpy(g[k]).cuda() for k in ("img", "gt_verb", "gt_nouns"))
pv, pn, pg = net(img, verb)
print("pred verbs", pv.argmax(1).tolist(), "ref", g["pred_verb"].argmax(1).tolist())
vl, nl = net.verb_loss(pv, verb), net.nouns_loss(pn, nouns)
(vl + nl).backward()
for k, p in net.named_parameters():
    if p.requires_grad:
        ref = g["grad/" + k]
        print("%-32s err %.3e  refmax %.3e" % (k, np.abs(p.grad.cpu().numpy() - ref).max(), np.abs(ref).max()))
ref = g["grad/role_emb.weight"]; got = net.role_emb.weight.grad.cpu().numpy()
print("row errs", np.abs(got-ref).max(1), "row refmax", np.abs(ref).max(1), "row gotmax", np.abs(got).max(1))
