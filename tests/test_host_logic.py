"""CPU tests of the host-side mirror of the reference interface: scorer (vectorised) against the reference's own
golden dictionaries and against the oracle on random cases; transforms; checkpoint helpers."""
import json
import os

import numpy as np
import torch

from golden_util import load, overfitting_json
from oracle.ref_encoder import RefEncoder, SyntheticEncoder
from oracle.ref_scorer import RefScorer
from situation_recognition_amd.imsitu_encoder import imsitu_encoder
from situation_recognition_amd.imsitu_scorer import imsitu_scorer


def test_scorer_matches_reference_goldens():
    g = load("g4_loss_scorer.npz")
    enc = imsitu_encoder(overfitting_json(), quiet=True)
    t = lambda k: torch.from_numpy(g[k])
    for name, keys in (("real", ("pred_verb", "pred_nouns", "gt_pred_nouns")), ("made", ("pv2", "pn2", "pg2"))):
        for k in (1, 5):
            sc = imsitu_scorer(enc, k, 3)
            sc.add_point_both(t(keys[0]), t("gt_verb"), t(keys[1]), t("gt_nouns"), t(keys[2]))
            res = sc.get_average_results_both()
            want = {kk.split("/")[-1]: float(v) for kk, v in g.items() if kk.startswith("score/%s/top%d/" % (name, k))}
            assert res.keys() == want.keys()
            for kk in want:
                assert abs(res[kk] - want[kk]) < 1e-12, (name, k, kk)
            assert len(sc.score_cards) == 5 and set(sc.score_cards[0]) == set(want)


def test_scorer_matches_oracle_on_random_full_size_vocabulary():
    V, NR, L, R, B = 40, 20, 60, 6, 64
    ora_enc = SyntheticEncoder(V, NR, L, R, seed=3)
    enc = imsitu_encoder.synthetic(V, NR, L, R, seed=3)
    assert torch.equal(enc.roles_to_verb_tensor_list, ora_enc.roles_to_verb_tensor_list)
    g = torch.Generator().manual_seed(0)
    verbs = torch.randint(0, V, (B,), generator=g)
    gold = torch.randint(0, L, (B, 3, R), generator=g)
    pv, pn, pg = torch.randn(B, V, generator=g), torch.randn(B, R, L, generator=g), torch.randn(B, R, L, generator=g)
    for b in range(B):                               # plant some hits so every criterion is exercised
        pv[b, verbs[b]] += 1.5
        for r in range(R):
            if (b + r) % 3 == 0:
                pn[b, r, gold[b, r % 3, r]] += 3.0
                pg[b, r, gold[b, (r + 1) % 3, r]] += 3.0
    for k in (1, 5):
        a, o = imsitu_scorer(enc, k, 3), RefScorer(ora_enc, k, 3)
        for lo in (0, 32):                            # two batches accumulate
            sl = slice(lo, lo + 32)
            a.add_point_both(pv[sl], verbs[sl], pn[sl], gold[sl], pg[sl])
            o.add_point_both(pv[sl], verbs[sl], pn[sl], gold[sl], pg[sl])
        ra, ro = a.get_average_results_both(), o.get_average_results_both()
        assert ra.keys() == ro.keys()
        for kk in ro:
            assert abs(ra[kk] - ro[kk]) < 1e-12, (k, kk, ra[kk], ro[kk])
        assert 0 < ro["value"] < 1


def test_transforms_shapes_and_normalisation():
    from PIL import Image
    rng = np.random.default_rng(0)
    img = Image.fromarray((rng.random((300, 420, 3)) * 255).astype(np.uint8))
    a, b = imsitu_encoder.train_transform(img), imsitu_encoder.dev_transform(img)
    assert a.shape == b.shape == (3, 224, 224) and a.dtype == torch.float32
    assert -2.2 < float(b.min()) and float(b.max()) < 2.7
    tall = Image.fromarray((rng.random((500, 230, 3)) * 255).astype(np.uint8))
    assert imsitu_encoder.dev_transform(tall).shape == (3, 224, 224)


def test_load_net_and_format_dict(tmp_path):
    from situation_recognition_amd import utils
    net = torch.nn.Linear(3, 2)
    other = torch.nn.Linear(3, 2)
    torch.save({"model_state_dict": other.state_dict(), "epoch": 3}, tmp_path / "ck")
    utils.load_net(str(tmp_path / "ck"), [net])
    assert torch.equal(net.weight, other.weight)
    assert utils.format_dict({"verb": 0.5, "value": 0.25}, "{:.2f}", "1-") == "1-verb: 50.00, 1-value: 25.00"
