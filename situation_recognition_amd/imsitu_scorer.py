"""The reference's metric (reference utils/imsitu_scorer.py) with the same class surface, computed for a whole
batch at once on the device the logits live on (the reference loops over samples, ranks and roles in Python and
synchronises on every tensor comparison; at batch 6144 that loop is the largest wall-clock item outside the model).

Criteria are the reference's, quirks included (imsitu_scorer.py:38-57): for the k-th ranked prediction the number of
(role, annotator) label hits is counted over the verb's real roles; "value" = any hit, "value-all" = the COUNT reaches the
number of roles; each card entry saturates at 1.
"""
import torch


class imsitu_scorer:
    KEYS1 = ("verb", "value", "value-all", "gt-value", "gt-value-all")
    KEYSK = ("verb", "value", "value-all")

    def __init__(self, encoder, topk, nref):
        self.encoder, self.topk, self.nref = encoder, topk, nref
        self._cards = []          # list of [B, nkeys] uint8 tensors (device)
        self._reduced = None      # (sums, total) over all ranks after all_reduce_()

    @property
    def keys(self):
        return self.KEYS1 if self.topk == 1 else self.KEYSK

    def _hits(self, ranked, gold, valid):
        # ranked [B,R,k] label ids; gold [B,3,R]; valid [B,R] -> [B,k] number of (role, annotator) matches
        eq = ranked[:, None, :, :] == gold[:, :, :, None]                 # [B,3,R,k]
        return (eq & valid[:, None, :, None]).sum(dim=(1, 2))

    def add_point_both(self, pred_verbs, verbs, pred_roles_nouns, roles_nouns, gt_pred_roles_nouns):
        dev = pred_verbs.device
        k = self.topk
        verbs = verbs.to(dev)
        gold = roles_nouns.to(dev)
        counts = self.encoder.role_counts.to(dev)[verbs]                   # [B]
        R = pred_roles_nouns.shape[1]
        valid = torch.arange(R, device=dev)[None, :] < counts[:, None]
        top_v = torch.topk(pred_verbs.float(), k, dim=1)[1]                # [B,k]
        top_n = torch.topk(pred_roles_nouns.float(), k, dim=2)[1]          # [B,R,k]
        hits = self._hits(top_n, gold, valid)                              # [B,k]
        cols = [(top_v == verbs[:, None]).any(1), (hits > 0).any(1), (hits >= counts[:, None]).any(1)]
        if k == 1:
            top_g = torch.topk(gt_pred_roles_nouns.float(), 1, dim=2)[1]
            gh = self._hits(top_g, gold, valid)[:, 0]
            cols += [gh > 0, gh >= counts]
        self._cards.append(torch.stack(cols, 1).to(torch.uint8))
        self._reduced = None

    @property
    def score_cards(self):
        """Per-sample cards in the reference's format (list of dicts); materialised on demand."""
        out = []
        for c in self._cards:
            for row in c.cpu().tolist():
                out.append({key: (1 if v else 0.0) for key, v in zip(self.keys, row)})
        return out

    def _totals(self):
        if self._cards:
            allc = torch.cat(self._cards, 0)
            return allc.sum(0, dtype=torch.int64).cpu().tolist(), allc.shape[0]
        return [0] * len(self.keys), 0

    def all_reduce_(self, group=None):
        """Data-parallel evaluation: add the other ranks' card sums and sample counts (exact integers), so that
        `get_average_results_both` returns the metric of the whole set on every rank."""
        import torch.distributed as dist
        sums, total = self._totals()
        t = torch.tensor(sums + [total], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=group)
        self._reduced = (t[:-1].tolist(), int(t[-1]))

    def get_average_results_both(self):
        sums, total = getattr(self, "_reduced", None) or self._totals()   # exact integer counts, then one division (as the reference)
        return {key: v / total for key, v in zip(self.keys, sums)}
