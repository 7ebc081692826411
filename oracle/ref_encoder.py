"""CPU restatement of the vocabulary / role-graph tables of the reference
(`utils/imsitu_encoder.py`).  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Pinned by tests/test_oracle_golden.py against goldens G1 (tables produced by the
reference's own class on its own fixture imSitu/overfitting.json).

Image transforms (imsitu_encoder.py:18-36) are data-loading, outside the hot
path, and are not restated.
"""
import torch


class RefEncoder:
    def __init__(self, train_set):
        """imsitu_encoder.py:39-63: ids are assigned in first-seen order while
        walking images -> frames -> (role, label) pairs."""
        self.verb_list, self.role_list, self.label_list = [], [], []
        self.roles_per_verb = {}
        self.max_label_count = 3
        for ann in train_set.values():
            v = ann["verb"]
            if v not in self.roles_per_verb:
                self.verb_list.append(v)
                self.roles_per_verb[v] = []
            for frame in ann["frames"]:
                for role, label in frame.items():
                    if role not in self.role_list:
                        self.role_list.append(role)
                    if role not in self.roles_per_verb[v]:
                        self.roles_per_verb[v].append(role)
                    if label not in self.label_list:
                        self.label_list.append(label)
        self.max_role_count = max(len(r) for r in self.roles_per_verb.values())
        # imsitu_encoder.py:71-89: per-verb role ids padded with num_roles
        pad = len(self.role_list)
        rows = []
        for v in self.verb_list:
            ids = [self.role_list.index(r) for r in self.roles_per_verb[v]]
            rows.append(ids + [pad] * (self.max_role_count - len(ids)))
        self.roles_to_verb_tensor_list = torch.tensor(rows, dtype=torch.int64)

    # imsitu_encoder.py:146-159
    def get_max_role_count(self): return self.max_role_count
    def get_num_verbs(self): return len(self.verb_list)
    def get_num_roles(self): return len(self.role_list)
    def get_num_labels(self): return len(self.label_list)
    def get_role_count(self, verb_id): return len(self.roles_per_verb[self.verb_list[int(verb_id)]])

    def get_role_ids_batch(self, verbs):
        """imsitu_encoder.py:172-180."""
        return self.roles_to_verb_tensor_list[torch.as_tensor(verbs).reshape(-1).cpu()]

    def get_adj_matrix_noself(self, verb_ids):
        """imsitu_encoder.py:209-229: with k real roles, A = e e^T (e = k ones then
        zeros), real diagonal cleared, padded diagonal set."""
        R = self.max_role_count
        out = torch.zeros(len(verb_ids), R, R)
        for b, v in enumerate(verb_ids):
            k = self.get_role_count(v)
            out[b, :k, :k] = 1.0
            out[b].fill_diagonal_(0.0)
            for p in range(k, R):
                out[b, p, p] = 1.0
        return out

    def get_label_ids(self, verb, frames):
        """imsitu_encoder.py:182-207: [n_frames, R] label ids in the verb's role
        order; unseen label -> id of 'UNK'; padded roles -> num_labels."""
        roles, L = self.roles_per_verb[verb], len(self.label_list)
        rows = []
        for fr in frames:
            ids = [self.label_list.index(fr[r]) if fr[r] in self.label_list
                   else self.label_list.index("UNK") for r in roles]
            rows.append(ids + [L] * (self.max_role_count - len(ids)))
        return torch.tensor(rows, dtype=torch.int64)

    def encode(self, item):
        """imsitu_encoder.py:161-166."""
        return self.verb_list.index(item["verb"]), self.get_label_ids(item["verb"], item["frames"])


class SyntheticEncoder(RefEncoder):
    """Vocabulary tables of a given size without a dataset (train.json is absent
    offline): V verbs, NR roles, L labels, max R roles per verb, role
    counts uniform in 1..R with verb 0 pinned at R."""

    def __init__(self, V=504, NR=190, L=2001, R=6, seed=1237):
        g = torch.Generator().manual_seed(seed)
        self.verb_list = ["v%d" % i for i in range(V)]
        self.role_list = ["r%d" % i for i in range(NR)]
        self.label_list = ["l%d" % i for i in range(L)]
        self.max_label_count, self.max_role_count = 3, R
        counts = torch.randint(1, R + 1, (V,), generator=g)
        counts[0] = R
        self.roles_per_verb, rows = {}, []
        for v in range(V):
            ids = torch.randperm(NR, generator=g)[: int(counts[v])].tolist()
            self.roles_per_verb[self.verb_list[v]] = [self.role_list[i] for i in ids]
            rows.append(ids + [NR] * (R - len(ids)))
        self.roles_to_verb_tensor_list = torch.tensor(rows, dtype=torch.int64)
