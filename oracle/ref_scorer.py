"""CPU restatement of the reference metric (`utils/imsitu_scorer.py:11-101`).
TEST INFRASTRUCTURE ONLY (oracle/__init__.py); pinned by golden G4.

The reference's criteria are kept as they are, quirks included: for the k-th
ranked prediction it counts label hits over (role, annotator) pairs, and calls
the frame "value-all" when that COUNT reaches the number of roles (not when every
role is hit), independent of whether the verb was right.
"""
import torch


class RefScorer:
    def __init__(self, encoder, topk, nref):
        self.encoder, self.topk, self.nref = encoder, topk, nref
        self.score_cards = []

    def _hits(self, ranked, gold, nroles, k):
        # ranked [R,topk] label ids, gold [3,R]
        return sum(int(ranked[r][k] == gold[n][r]) for r in range(nroles) for n in range(3))

    def add_point_both(self, pred_verbs, verbs, pred_roles_nouns, roles_nouns, gt_pred_roles_nouns):
        keys = ["verb", "value", "value-all"] + (["gt-value", "gt-value-all"] if self.topk == 1 else [])
        for i in range(verbs.shape[0]):
            card = dict.fromkeys(keys, 0.0)
            verb, gold = verbs[i], roles_nouns[i]
            top_v = torch.topk(pred_verbs[i], self.topk)[1]
            top_n = torch.topk(pred_roles_nouns[i], self.topk)[1]
            nroles = self.encoder.get_role_count(verb)
            for k in range(self.topk):
                if top_v[k] == verb:
                    card["verb"] += 1
                hits = self._hits(top_n, gold, nroles, k)
                card["value-all"] += hits >= nroles
                card["value"] += hits > 0
            if self.topk == 1:
                top_g = torch.topk(gt_pred_roles_nouns[i], 1)[1]
                hits = self._hits(top_g, gold, nroles, 0)
                card["gt-value-all"] += hits >= nroles
                card["gt-value"] += hits > 0
            self.score_cards.append({k: (1 if v > 0 else v) for k, v in card.items()})

    def get_average_results_both(self):
        n = len(self.score_cards)
        keys = self.score_cards[0].keys()
        return {k: sum(c[k] for c in self.score_cards) / n for k in keys}
