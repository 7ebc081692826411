"""Pins the CPU oracle (oracle/) to the reference: every fixture here was written
by oracle/gen_golden.py by running the reference's own code (reference model.py,
utils/imsitu_encoder.py, utils/imsitu_scorer.py) on CPU fp32."""
import numpy as np
import pytest
import torch

from golden_util import load, oracle_fcggnn, overfitting_json, sub
from oracle.ref_encoder import RefEncoder
from oracle.ref_model import RefGGSNN, train_step
from oracle.ref_scorer import RefScorer


def test_g1_encoder_tables():
    g = load("g1_encoder.npz")
    ts = overfitting_json()
    enc = RefEncoder(ts)
    assert enc.verb_list == list(g["verb_list"])
    assert enc.role_list == list(g["role_list"])
    assert enc.label_list == list(g["label_list"])
    assert enc.get_max_role_count() == int(g["max_role_count"]) == 4
    assert np.array_equal(enc.roles_to_verb_tensor_list.numpy(), g["roles_to_verb"])
    V = enc.get_num_verbs()
    assert [enc.get_role_count(v) for v in range(V)] == list(g["role_counts"])
    assert np.array_equal(enc.get_adj_matrix_noself(torch.arange(V)).numpy(), g["adj_all_verbs"])
    assert np.array_equal(enc.get_role_ids_batch(torch.tensor([4, 0, 2, 2, 1])).numpy(), g["role_ids_batch"])
    for i, ann in enumerate(ts.values()):
        v, lab = enc.encode(ann)
        assert v == int(g["encode_verb_%d" % i])
        assert np.array_equal(lab.numpy(), g["encode_labels_%d" % i])


def _ggsnn(g):
    gg = RefGGSNN(64, steps=4)
    gg.load_state_dict(sub(g, "w/"))
    return gg


@pytest.mark.parametrize("path", ["noun", "verb"])
def test_g2_ggsnn_forward_backward(path):
    g = load("g2_ggsnn.npz")
    gg = _ggsnn(g)
    if path == "noun":
        h = torch.from_numpy(g["h0"]).requires_grad_(True)
        out = gg(h, mask=torch.from_numpy(g["mask"]), verb=False)
        c, want, gp, gh = g["cn"], g["out_n"], "gn/", "gn/h0"
    else:
        h = torch.from_numpy(g["hv"]).requires_grad_(True)
        out = gg(h, mask=None, verb=True)
        c, want, gp, gh = g["cv"], g["out_v"], "gv/", "gv/hv"
    assert np.abs(out.detach().numpy() - want).max() < 2e-6
    (out * torch.from_numpy(c)).sum().backward()
    assert np.abs(h.grad.numpy() - g[gh]).max() < 1e-5
    for k, p in gg.named_parameters():
        ref = g[gp + k]
        assert np.abs(p.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k


def test_g2_algebraic_neighbour_form_is_the_same_quantity():
    """(A.H) W_p^T + R b_p  ==  sum_j [W_p(A_ij h_j) + b_p]   """
    g = load("g2_ggsnn.npz")
    gg = _ggsnn(g)
    h, m = torch.from_numpy(g["h0"]), torch.from_numpy(g["mask"])
    with torch.no_grad():
        a = gg.neighbours(h, m, False)
        b = gg.neighbours_algebraic(h, m)
    assert (a - b).abs().max() < 2e-5


@pytest.mark.parametrize("tag", ["bottleneck", "basic"])
def test_g3_fcggnn_eval_end_to_end(tag):
    g = load("g3_fcggnn_%s.npz" % tag)
    net, enc, _ = oracle_fcggnn(g)
    net.eval()
    img, gv = torch.from_numpy(g["img"]), torch.from_numpy(g["gt_verb"])
    with torch.no_grad():
        pv, pn, pg = net(img, gv)
        given = net.predict_nouns(img, torch.from_numpy(g["given_verbs"]), 5)
        assert (net.convnet_verbs(img) - torch.from_numpy(g["feat_verbs"])).abs().max() < 1e-5
    assert np.abs(pv.numpy() - g["pred_verb"]).max() < 1e-5
    assert np.abs(pn.numpy() - g["pred_nouns"]).max() < 1e-5
    assert np.abs(pg.numpy() - g["gt_pred_nouns"]).max() < 1e-5
    assert np.abs(given.numpy() - g["pred_nouns_given"]).max() < 1e-5


def test_g4_losses_and_scorer():
    g = load("g4_loss_scorer.npz")
    g3 = load("g3_fcggnn_bottleneck.npz")
    net, enc, _ = oracle_fcggnn(g3)
    t = lambda k: torch.from_numpy(g[k])
    assert abs(float(net.verb_loss(t("pred_verb"), t("gt_verb"))) - float(g["verb_loss"])) < 1e-6
    assert abs(float(net.nouns_loss(t("pred_nouns"), t("gt_nouns"))) - float(g["nouns_loss"])) < 1e-5
    assert abs(float(net.nouns_loss(t("gt_pred_nouns"), t("gt_nouns"))) - float(g["gt_nouns_loss"])) < 1e-5
    for name, keys in (("real", ("pred_verb", "pred_nouns", "gt_pred_nouns")), ("made", ("pv2", "pn2", "pg2"))):
        for k in (1, 5):
            sc = RefScorer(enc, k, 3)
            sc.add_point_both(t(keys[0]), t("gt_verb"), t(keys[1]), t("gt_nouns"), t(keys[2]))
            res = sc.get_average_results_both()
            want = {kk.split("/")[-1]: float(v) for kk, v in g.items() if kk.startswith("score/%s/top%d/" % (name, k))}
            assert set(res) == set(want)
            for kk in want:
                assert abs(res[kk] - want[kk]) < 1e-12, (name, k, kk)
    # the constructed case must not be degenerate
    assert 0 < float(g["score/made/top1/value"]) and float(g["score/made/top1/verb"]) < 1


def test_g5_one_training_step():
    g = load("g5_train_step.npz")
    net, enc, _ = oracle_fcggnn(load("g3_fcggnn_basic.npz"))
    net.train()
    net.verb_classifier[0].p = 0.0
    net.nouns_classifier[0].p = 0.0
    opt = torch.optim.Adamax([p for p in net.parameters() if p.requires_grad], lr=0.002)
    img, verb, nouns = torch.from_numpy(g["img"]), torch.from_numpy(g["gt_verb"]), torch.from_numpy(g["gt_nouns"])
    # unclipped gradients first (a separate backward on a copy of the net)
    import copy
    twin = copy.deepcopy(net)
    pv, pn, _ = twin(img, verb)
    (twin.verb_loss(pv, verb) + twin.nouns_loss(pn, nouns)).backward()
    for k, p in twin.named_parameters():
        if p.requires_grad:
            ref = g["grad/" + k]
            assert np.abs(p.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k
    r = train_step(net, opt, img, verb, nouns)
    assert abs(float(r["verb_loss"]) - float(g["verb_loss"])) < 1e-5
    assert abs(float(r["nouns_loss"]) - float(g["nouns_loss"])) < 1e-4
    assert abs(float(r["gt_nouns_loss"]) - float(g["gt_nouns_loss"])) < 1e-4
    assert abs(float(r["grad_norm"]) - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
    after = sub(g, "after/")
    for k, v in net.state_dict().items():
        assert (v.float() - after[k].float()).abs().max() <= 2e-5 * max(1.0, float(after[k].float().abs().max())), k
    # the gt-verb branch must not have contributed gradient; frozen backbone has none
    assert not any(k.startswith("grad/convnet") for k in g)
