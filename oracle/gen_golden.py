#!/usr/bin/env python3
"""Writes tests/golden/*.npz by RUNNING THE REFERENCE ITSELF on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the authoring container only (it reads
/root/reference by path; on the GPU box that path does not exist and the
committed fixtures are used instead):

    python oracle/gen_golden.py            # -> tests/golden/g1..g5 *.npz

How the reference is imported: `model.py` and `utils/imsitu_encoder.py` do
`import torchvision as tv` at module level and torchvision is not installed in
this image.  A `types.ModuleType("torchvision")` is registered whose
`transforms.*` are inert and whose `models.resnet152(pretrained, progress)`
returns oracle.ref_resnet.RefResNet (our restatement of the published graph; the
ImageNet weights behind `pretrained=True` are a network fetch and unavailable).
Everything else -- GGSNN, FCGGNN.predict_*/forward, the losses, the encoder tables,
the scorer -- is the reference's own code executing.  Goldens therefore pin the
oracle for A3-A13 of SURVEY 8(a); the backbone arithmetic stays "parity unpinned"
(see oracle/ref_resnet.py).
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle.ref_resnet import RefResNet, perturb_batchnorm_  # noqa: E402

_BACKBONE_CFG = {}


def _install_torchvision_standin():
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.models = types.ModuleType("torchvision.models")

    class _Inert:
        def __init__(self, *a, **k):
            pass

    for n in ("Normalize", "Compose", "Resize", "RandomCrop", "RandomHorizontalFlip",
              "ToTensor", "CenterCrop"):
        setattr(tv.transforms, n, _Inert)
    tv.models.resnet152 = lambda pretrained=True, progress=False: RefResNet(**_BACKBONE_CFG)
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms
    sys.modules["torchvision.models"] = tv.models


def _np(t):
    return t.detach().cpu().numpy().copy()      # copy: later in-place ops (grad clipping) must not alias


def _state(prefix, module):
    return {prefix + k: _np(v) for k, v in module.state_dict().items()}


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing to do (fixtures are committed)")
        return
    warnings.filterwarnings("ignore")
    _install_torchvision_standin()
    sys.path.insert(0, REF)
    import model as ref_model                                    # noqa: E402
    from utils import imsitu_encoder as ref_enc_mod              # noqa: E402
    from utils import imsitu_scorer as ref_scorer_mod            # noqa: E402
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)

    train_set = json.load(open(os.path.join(REF, "imSitu", "overfitting.json")))
    devnull = open(os.devnull, "w")
    so, sys.stdout = sys.stdout, devnull
    enc = ref_enc_mod.imsitu_encoder(train_set)
    sys.stdout = so

    # ---- G1: encoder tables on the reference's own fixture -------------------
    V = enc.get_num_verbs()
    g1 = dict(
        verb_list=np.array(enc.verb_list), role_list=np.array(enc.role_list),
        label_list=np.array(enc.label_list),
        max_role_count=np.int64(enc.get_max_role_count()),
        roles_to_verb=_np(enc.roles_to_verb_tensor_list),
        role_counts=np.array([enc.get_role_count(v) for v in range(V)]),
        adj_all_verbs=_np(enc.get_adj_matrix_noself(torch.arange(V))),
        role_ids_batch=_np(enc.get_role_ids_batch(torch.tensor([4, 0, 2, 2, 1]))),
    )
    for i, (name, ann) in enumerate(train_set.items()):
        v, lab = enc.encode(ann)
        g1["encode_verb_%d" % i] = np.int64(v)
        g1["encode_labels_%d" % i] = _np(lab)
    np.savez_compressed(os.path.join(OUT, "g1_encoder.npz"), **g1)

    # ---- G2: GGSNN forward + gradients (D=64; R=6 masks; verb path) -----------
    torch.manual_seed(20)
    D, R, B = 64, 6, 7
    gg = ref_model.GGSNN(D)
    with torch.no_grad():                       # larger weights => gates leave the linear regime
        for p in gg.parameters():
            p.mul_(3.0)
    counts = [6, 1, 3, 5, 2, 4, 6]
    mask = torch.zeros(B, R, R)
    for b, k in enumerate(counts):              # same rule as imsitu_encoder.py:209-229
        mask[b, :k, :k] = 1
        mask[b].fill_diagonal_(0)
        for p_ in range(k, R):
            mask[b, p_, p_] = 1
    h0 = torch.randn(B * R, D).relu_().requires_grad_(True)
    hv = torch.randn(B, D).relu_().requires_grad_(True)
    cn, cv = torch.randn(B * R, D), torch.randn(B, D)
    out_n = gg(h0, mask=mask, verb=False)
    out_v = gg(hv, mask=None, verb=True)
    g2 = dict(mask=_np(mask), h0=_np(h0), hv=_np(hv), cn=_np(cn), cv=_np(cv),
              out_n=_np(out_n), out_v=_np(out_v))
    g2.update(_state("w/", gg))
    (out_n * cn).sum().backward()
    g2["gn/h0"] = _np(h0.grad)
    g2.update({"gn/" + k: _np(p.grad) for k, p in gg.named_parameters()})
    gg.zero_grad()
    (out_v * cv).sum().backward()
    g2["gv/hv"] = _np(hv.grad)
    g2.update({"gv/" + k: _np(p.grad) for k, p in gg.named_parameters()})
    np.savez_compressed(os.path.join(OUT, "g2_ggsnn.npz"), **g2)

    # ---- G3: FCGGNN eval-mode end to end through the stand-in backbone --------
    for tag, cfg, res in (("bottleneck", dict(depth=50, width=8, blocks=(1, 2, 1, 1)), 64),
                          ("basic", dict(depth=18, width=8, blocks=(1, 1, 2, 1)), 64)):
        _BACKBONE_CFG.clear()
        _BACKBONE_CFG.update(cfg)
        torch.manual_seed(30)
        Dh = RefResNet(**cfg).fc.in_features
        net = ref_model.FCGGNN(enc, Dh)
        perturb_batchnorm_(net.convnet_verbs, 31)
        perturb_batchnorm_(net.convnet_nouns, 32)
        with torch.no_grad():                   # keep features/logits O(1) and argmax well separated
            net.verb_emb.weight.mul_(1.5)
            net.role_emb.weight.mul_(1.5)
        net.eval()
        img = torch.randn(5, 3, res, res).clamp_(-2.2, 2.7)
        gt_verb = torch.tensor([0, 1, 2, 3, 4])
        with torch.no_grad():
            pv, pn, pg = net(img, gt_verb)
            fv = net.convnet_verbs(img)
            fn = net.convnet_nouns(img)
            pn_given = net.predict_nouns(img, torch.tensor([1, 1, 0, 4, 2]), 5)
        g3 = dict(img=_np(img), gt_verb=_np(gt_verb), pred_verb=_np(pv), pred_nouns=_np(pn),
                  gt_pred_nouns=_np(pg), feat_verbs=_np(fv), feat_nouns=_np(fn),
                  given_verbs=np.array([1, 1, 0, 4, 2]), pred_nouns_given=_np(pn_given),
                  cfg_depth=np.int64(cfg["depth"]), cfg_width=np.int64(cfg["width"]),
                  cfg_blocks=np.array(cfg["blocks"]), D=np.int64(Dh))
        g3.update(_state("state/", net))
        np.savez_compressed(os.path.join(OUT, "g3_fcggnn_%s.npz" % tag), **g3)

        # ---- G4 (once): losses + scorer on those logits ---------------------
        if tag == "bottleneck":
            gt_nouns = torch.stack([enc.encode(a)[1] for a in train_set.values()])     # [5,3,R]
            vl = net.verb_loss(pv, gt_verb)
            nl = net.nouns_loss(pn, gt_nouns)
            gl = net.nouns_loss(pg, gt_nouns)
            g4 = dict(pred_verb=_np(pv), pred_nouns=_np(pn), gt_pred_nouns=_np(pg),
                      gt_verb=_np(gt_verb), gt_nouns=_np(gt_nouns),
                      verb_loss=_np(vl), nouns_loss=_np(nl), gt_nouns_loss=_np(gl))
            # also logits constructed to hit/miss specific criteria
            torch.manual_seed(40)
            pv2 = torch.randn(5, V)
            pn2 = torch.randn(5, enc.get_max_role_count(), enc.get_num_labels())
            pg2 = torch.randn_like(pn2)
            for b in range(5):
                for r in range(enc.get_role_count(gt_verb[b])):
                    if (b + r) % 2 == 0:
                        pn2[b, r, gt_nouns[b, r % 3, r]] += 6.0
                    if b % 2 == 1:
                        pg2[b, r, gt_nouns[b, 0, r]] += 6.0
                if b < 3:
                    pv2[b, gt_verb[b]] += 5.0
            g4.update(pv2=_np(pv2), pn2=_np(pn2), pg2=_np(pg2))
            for name, (a, b_, c) in (("real", (pv, pn, pg)), ("made", (pv2, pn2, pg2))):
                for k in (1, 5):
                    sc = ref_scorer_mod.imsitu_scorer(enc, k, 3)
                    sc.add_point_both(a, gt_verb, b_, gt_nouns, c)
                    for key, val in sc.get_average_results_both().items():
                        g4["score/%s/top%d/%s" % (name, k, key)] = np.float64(val)
            np.savez_compressed(os.path.join(OUT, "g4_loss_scorer.npz"), **g4)

        # ---- G5 (once, small model): one full training step, sr.py:63-83 -----
        if tag == "basic":
            net.train()
            opt = torch.optim.Adamax(filter(lambda p: p.requires_grad, net.parameters()), lr=0.002)
            gt_nouns = torch.stack([enc.encode(a)[1] for a in train_set.values()])
            # Dropout(0.5) is stochastic: the step is made reproducible by running with
            # p=0 (identity).  The masked path is tested HIP-vs-oracle with injected masks.
            net.verb_classifier[0].p = 0.0
            net.nouns_classifier[0].p = 0.0
            opt.zero_grad()
            pv, pn, pg = net(img, gt_verb)
            vl, nl, gl = net.verb_loss(pv, gt_verb), net.nouns_loss(pn, gt_nouns), net.nouns_loss(pg, gt_nouns)
            (vl + nl).backward()
            grads = {"grad/" + k: _np(p.grad) for k, p in net.named_parameters() if p.grad is not None}
            gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 1)
            opt.step()
            g5 = dict(img=_np(img), gt_verb=_np(gt_verb), gt_nouns=_np(gt_nouns),
                      pred_verb=_np(pv), pred_nouns=_np(pn), gt_pred_nouns=_np(pg),
                      verb_loss=_np(vl), nouns_loss=_np(nl), gt_nouns_loss=_np(gl), grad_norm=_np(gn))
            g5.update(grads)
            g5.update(_state("after/", net))
            np.savez_compressed(os.path.join(OUT, "g5_train_step.npz"), **g5)

    for f in sorted(os.listdir(OUT)):
        print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
