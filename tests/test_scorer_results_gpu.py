"""SURVEY 8(f) rows 1 and 4 on the device:
  * the vectorised scorer fed DEVICE tensors (as sr.train / sr.eval feed it) against the reference's own G4 dictionaries and,
    at a full-size vocabulary and batch 4096, against the per-sample oracle on a subset;
  * single-image inference `sr.results` (reference sr.py:235-281, incl. its softmax over ROLES, dim=0, at line 264) against the
    CPU oracle's logits post-processed the same way, and eager vs hipGraph-replayed backbones bit for bit.
"""
import json

import numpy as np
import pytest
import torch

from golden_util import load, overfitting_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import ops
    ops.lib()
    return torch.device("cuda")


def test_scorer_on_device_matches_reference_goldens(gpu):
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.imsitu_scorer import imsitu_scorer
    g = load("g4_loss_scorer.npz")
    enc = imsitu_encoder(overfitting_json(), quiet=True)
    t = lambda k: torch.from_numpy(g[k]).to(gpu)
    for name, keys in (("real", ("pred_verb", "pred_nouns", "gt_pred_nouns")), ("made", ("pv2", "pn2", "pg2"))):
        for k in (1, 5):
            sc = imsitu_scorer(enc, k, 3)
            sc.add_point_both(t(keys[0]), t("gt_verb"), t(keys[1]), t("gt_nouns"), t(keys[2]))
            assert sc._cards[0].is_cuda
            res = sc.get_average_results_both()
            want = {kk.split("/")[-1]: float(v) for kk, v in g.items() if kk.startswith("score/%s/top%d/" % (name, k))}
            assert res.keys() == want.keys()
            for kk in want:
                assert abs(res[kk] - want[kk]) < 1e-12, (name, k, kk)


def test_scorer_on_device_batch4096_matches_oracle_subset(gpu):
    """imSitu-sized vocabulary, bf16 noun logits as the model emits them in the benchmark configuration, batch 4096 in two
    calls; the per-sample oracle (a Python triple loop) scores a 384-sample subset, whose cards must equal the device's."""
    from oracle.ref_encoder import SyntheticEncoder
    from oracle.ref_scorer import RefScorer
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.imsitu_scorer import imsitu_scorer
    enc, ora_enc = imsitu_encoder.synthetic(), SyntheticEncoder()
    V, L, R, B = 504, 2001, 6, 4096
    g = torch.Generator().manual_seed(11)
    verbs = torch.randint(0, V, (B,), generator=g)
    gold = torch.randint(0, L, (B, 3, R), generator=g)
    counts = enc.role_counts[verbs]
    gold[(torch.arange(R)[None, :] >= counts[:, None])[:, None, :].expand(B, 3, R)] = L
    pv = torch.randn(B, V, generator=g)
    pn, pg = torch.randn(B, R, L, generator=g), torch.randn(B, R, L, generator=g)
    idx = torch.arange(B)
    pv[idx[::2], verbs[::2]] += 4.0                              # half of the verbs right
    for r in range(R):                                           # plant label hits so every criterion fires somewhere
        sel = idx[(idx + r) % 3 != 0]
        lab = gold[sel, r % 3, r].clamp(max=L - 1)
        pn[sel, r, lab] += 6.0
        pg[sel, r, gold[sel, (r + 1) % 3, r].clamp(max=L - 1)] += 6.0
    pn, pg = pn.bfloat16(), pg.bfloat16()
    sub = torch.arange(0, B, B // 384)[:384]
    for k in (1, 5):
        dev_sc, ora = imsitu_scorer(enc, k, 3), RefScorer(ora_enc, k, 3)
        for lo in (0, B // 2):
            sl = slice(lo, lo + B // 2)
            dev_sc.add_point_both(pv[sl].to(gpu), verbs[sl].to(gpu), pn[sl].to(gpu), gold[sl].to(gpu), pg[sl].to(gpu))
        ora.add_point_both(pv[sub], verbs[sub], pn[sub].float(), gold[sub], pg[sub].float())
        cards = dev_sc.score_cards
        assert len(cards) == B
        for j, i in enumerate(sub.tolist()):
            assert cards[i] == ora.score_cards[j], (k, i, cards[i], ora.score_cards[j])
        res = dev_sc.get_average_results_both()
        assert 0.4 < res["verb"] < 0.7 and 0.0 < res["value-all"] < 1.0 and 0.0 < res["value"] <= 1.0
        sums = {key: sum(c[key] for c in cards) / B for key in res}
        for key in res:
            assert abs(res[key] - sums[key]) < 1e-12


def _space_json(tmp_path, enc, ann):
    """Synthetic imsitu_space.json of the reference's shape ({"nouns": {id: {"gloss": [..]}}, "verbs": {v: {"roles": {..}}}});
    the real file is absent from the reference checkout (.MISSING_LARGE_BLOBS)."""
    nouns = {lab: {"gloss": ["gloss-of-%s" % lab]} for lab in enc.label_list if lab not in ("", "UNK")}
    verbs = {v: {"roles": {r: {} for r in enc.roles_per_verb[v]}} for v in enc.verb_list}
    path = tmp_path / "imsitu_space.json"
    json.dump({"nouns": nouns, "verbs": verbs}, open(path, "w"))
    return str(path)


def test_results_single_image_vs_oracle_and_graph_replay(gpu, tmp_path, capsys):
    from PIL import Image
    from oracle.ref_encoder import RefEncoder
    from oracle.ref_model import RefBackbone, RefFCGGNN
    from situation_recognition_amd import sr
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.model import FCGGNN
    ann = overfitting_json()
    enc = imsitu_encoder(ann, quiet=True)
    space = _space_json(tmp_path, enc, ann)
    rng = np.random.default_rng(3)
    img_path = str(tmp_path / "x.jpg")
    Image.fromarray((rng.random((260, 340, 3)) * 255).astype(np.uint8)).save(img_path)
    cfg = dict(depth=18, width=16, blocks=(1, 1, 1, 1))
    torch.manual_seed(2)
    ora = RefFCGGNN(RefEncoder(ann), 128, steps=4, backbone_factory=lambda: RefBackbone(**cfg)).eval()
    net = FCGGNN(enc, 128, steps=4, backbone=18, dtype=torch.float32, width=16, blocks=(1, 1, 1, 1))
    net.load_state_dict(ora.state_dict(), strict=True)
    net.cuda()

    # expectation: the oracle's logits, post-processed as reference sr.py:255-279 does
    x = enc.dev_transform(Image.open(img_path).convert("RGB")).unsqueeze(0)
    with torch.no_grad():
        lv = ora.predict_verb(x, 1)
        vt = torch.argmax(lv, 1)
        vprob = torch.max(torch.softmax(lv, dim=1)).item() * 100
        ln = ora.predict_nouns(x, vt, 1).squeeze(0)
    nt = torch.argmax(ln, 1)
    lprob = [p.item() * 100 for p in torch.max(torch.softmax(ln, dim=0), 1)[0]]       # sr.py:264: softmax over ROLES
    roles = enc.roles_per_verb[enc.verb_list[int(vt)]]
    want_labels = {}
    for c, i in enumerate(nt[: len(roles)].tolist()):
        lab = enc.label_list[i]
        want_labels[roles[c]] = "-" if lab in ("", "UNK") else "gloss-of-%s" % lab

    def check(out):
        verb_name, verb_prob, labels, labels_prob = out
        assert verb_name == enc.verb_list[int(vt)]
        assert abs(verb_prob - vprob) < 1e-2
        assert labels == want_labels
        assert len(labels_prob) == ln.shape[0] and max(abs(a - b) for a, b in zip(labels_prob, lprob)) < 1e-2

    eager = sr.results(net, img_path, enc, "", space_json=space)
    assert "No ground truth verb found" in capsys.readouterr().out
    check(eager)
    # given verb: no verb inference, probability 100 (sr.py:249-251)
    given = sr.results(net, img_path, enc, enc.verb_list[3], space_json=space)
    assert given[0] == enc.verb_list[3] and given[1] == 100 and set(given[2]) == set(enc.roles_per_verb[enc.verb_list[3]])
    # the same through captured hipGraphs: bit-identical numbers, also on replay
    net.enable_graphs()
    for _ in range(2):
        g = sr.results(net, img_path, enc, "", space_json=space)
        assert g[0] == eager[0] and g[1] == eager[1] and g[2] == eager[2] and g[3] == eager[3]
    assert len(net.convnet_verbs._graphs) == 1 and len(net.convnet_nouns._graphs) == 1
    # an in-place edit of ANY backbone weight must not replay a graph with stale folded packs
    with torch.no_grad():
        net.convnet_nouns.model.layer2[0].conv1.weight.mul_(1.5)
        ora.convnet_nouns.model.layer2[0].conv1.weight.mul_(1.5)
        ln2 = ora.predict_nouns(x, vt, 1).squeeze(0)
    g2 = sr.results(net, img_path, enc, "", space_json=space)
    lprob2 = [p.item() * 100 for p in torch.max(torch.softmax(ln2, dim=0), 1)[0]]
    assert max(abs(a - b) for a, b in zip(g2[3], lprob2)) < 1e-2
    assert max(abs(a - b) for a, b in zip(g2[3], eager[3])) > 1e-6
    net.enable_graphs(False)
    e2 = sr.results(net, img_path, enc, "", space_json=space)
    assert e2[3] == g2[3]


def test_graph_replayed_features_equal_eager_uint8_and_fp32(gpu):
    from situation_recognition_amd.model import resnet
    torch.manual_seed(4)
    net = resnet(None, depth=50, width=16, dtype=torch.bfloat16).cuda().eval()
    gen = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (2, 96, 96, 3), generator=gen, dtype=torch.uint8).cuda()
    f32 = torch.randn(2, 3, 96, 96, generator=gen).cuda()
    with torch.no_grad():
        e_u8, e_f = net(u8).clone(), net(f32).clone()
        net.use_graphs = True
        for _ in range(2):
            assert torch.equal(net(u8), e_u8) and torch.equal(net(f32), e_f)
        assert len(net._graphs) == 2
        # load_state_dict invalidates the captured graphs and drops pending num_batches_tracked increments
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd["model.layer1.0.conv1.weight"] = sd["model.layer1.0.conv1.weight"] * 0.5
        net.load_state_dict(sd)
        assert len(net._graphs) == 0
        out = net(f32)
        net.use_graphs = False
        assert torch.equal(out, net(f32)) and not torch.equal(out, e_f)


def test_num_batches_tracked_not_double_counted_after_load(gpu):
    from situation_recognition_amd.model import resnet
    torch.manual_seed(6)
    net = resnet(None, depth=18, width=16, blocks=(1, 1, 1, 1), dtype=torch.float32).cuda().train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    net(x); net(x)                                              # two pending increments, not yet flushed
    sd0 = {k: v.clone() for k, v in resnet(None, depth=18, width=16, blocks=(1, 1, 1, 1), dtype=torch.float32).state_dict().items()}
    sd0["model.bn1.num_batches_tracked"] = torch.tensor(7)
    net.load_state_dict(sd0, strict=True)
    assert int(net.state_dict()["model.bn1.num_batches_tracked"]) == 7
    net(x)
    assert int(net.state_dict()["model.bn1.num_batches_tracked"]) == 8
