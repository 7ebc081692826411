"""Vocabulary / role-graph tables with the reference's `imsitu_encoder` surface
(reference utils/imsitu_encoder.py), rebuilt around flat lookup tables so the hot path can gather
them on the GPU by verb id instead of looping over the batch in Python
(reference imsitu_encoder.py:172-180 and 209-229 are per-sample Python loops).

Same ids as the reference: verbs, roles and labels are numbered in first-seen order while walking
images -> frames -> (role, label); padded role slots hold `num_roles`, padded label slots `num_labels`.
"""
import random

import torch

_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _to_normalised_tensor(img):
    """ToTensor + Normalize(ImageNet mean/std) of a PIL RGB image -> float32 [3,H,W]."""
    import numpy as np
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    return (a - torch.tensor(_MEAN).view(3, 1, 1)) / torch.tensor(_STD).view(3, 1, 1)


def _resize_shorter(img, size):
    from PIL import Image
    w, h = img.size
    if w <= h:
        nw, nh = size, max(size, int(round(h * size / w)))
    else:
        nw, nh = max(size, int(round(w * size / h))), size
    return img.resize((nw, nh), Image.BILINEAR)


def train_transform(img):
    """reference imsitu_encoder.py:21-29: Resize(224) -> RandomCrop(224) -> RandomHorizontalFlip -> ToTensor -> Normalize
    (PIL-only restatement: torchvision is not a dependency here)."""
    from PIL import Image
    img = _resize_shorter(img, 224)
    w, h = img.size
    x0, y0 = random.randint(0, w - 224), random.randint(0, h - 224)
    img = img.crop((x0, y0, x0 + 224, y0 + 224))
    if random.random() < 0.5:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    return _to_normalised_tensor(img)


def dev_transform(img):
    """reference imsitu_encoder.py:31-36: Resize(224) -> CenterCrop(224) -> ToTensor -> Normalize."""
    img = _resize_shorter(img, 224)
    w, h = img.size
    x0, y0 = int(round((w - 224) / 2.0)), int(round((h - 224) / 2.0))
    return _to_normalised_tensor(img.crop((x0, y0, x0 + 224, y0 + 224)))


class imsitu_encoder:
    train_transform = staticmethod(train_transform)
    dev_transform = staticmethod(dev_transform)

    def __init__(self, train_set=None, quiet=False):
        self.max_label_count = 3
        self.verb_list, self.role_list, self.label_list = [], [], []
        self.roles_per_verb = {}
        self._device_tables = {}
        if train_set is None:
            return
        verb_id, role_id, label_id = {}, {}, {}
        for ann in train_set.values():                     # imsitu_encoder.py:39-63
            v = ann["verb"]
            if v not in verb_id:
                verb_id[v] = len(self.verb_list)
                self.verb_list.append(v)
                self.roles_per_verb[v] = []
            mine = self.roles_per_verb[v]
            for frame in ann["frames"]:
                for role, label in frame.items():
                    if role not in role_id:
                        role_id[role] = len(self.role_list)
                        self.role_list.append(role)
                    if role not in mine:
                        mine.append(role)
                    if label not in label_id:
                        label_id[label] = len(self.label_list)
                        self.label_list.append(label)
        self._finish(quiet)

    # ------------------------------------------------------------------ tables
    def _finish(self, quiet=True):
        self._verb_id = {v: i for i, v in enumerate(self.verb_list)}
        self._role_id = {r: i for i, r in enumerate(self.role_list)}
        self._label_id = {l: i for i, l in enumerate(self.label_list)}
        self.max_role_count = max((len(r) for r in self.roles_per_verb.values()), default=0)
        V, R, NR = len(self.verb_list), self.max_role_count, len(self.role_list)
        table = torch.full((V, R), NR, dtype=torch.int64)              # imsitu_encoder.py:71-89
        counts = torch.zeros(V, dtype=torch.int64)
        for i, v in enumerate(self.verb_list):
            ids = [self._role_id[r] for r in self.roles_per_verb[v]]
            table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int64)
            counts[i] = len(ids)
        self.roles_to_verb_tensor_list = table
        self.role_counts = counts
        # imsitu_encoder.py:93-112 (0/1 presence) and 209-229 (adjacency without self loops on
        # real roles, self loops on padded slots), for every verb at once
        present = (torch.arange(R)[None, :] < counts[:, None]).float()             # [V,R]
        self.verb2role_encoding = [present[i].long() for i in range(V)]
        eye = torch.eye(R)
        self.adj_table = present[:, :, None] * present[:, None, :] * (1 - eye) + (1 - present)[:, :, None] * eye
        self._device_tables = {}
        if not quiet:
            print('train set stats: \n\t verb count:', V, '\n\t role count:', NR,
                  '\n\t label count:', len(self.label_list), '\n\t max role count:', R)

    @classmethod
    def synthetic(cls, V=504, NR=190, L=2001, R=6, seed=1237):
        """Tables of the full imSitu size without the dataset (train.json is not shipped with the
        reference): role counts uniform in 1..R, verb 0 pinned at R."""
        g = torch.Generator().manual_seed(seed)
        e = cls()
        e.verb_list = ["v%d" % i for i in range(V)]
        e.role_list = ["r%d" % i for i in range(NR)]
        e.label_list = ["l%d" % i for i in range(L)]
        counts = torch.randint(1, R + 1, (V,), generator=g)
        counts[0] = R
        for v in range(V):
            ids = torch.randperm(NR, generator=g)[: int(counts[v])].tolist()
            e.roles_per_verb[e.verb_list[v]] = [e.role_list[i] for i in ids]
        e._finish()
        return e

    def device_tables(self, device):
        """(role_table int32 [V,R], adj_table fp32 [V,R,R], role_counts int64 [V]) resident on `device`."""
        key = str(device)
        if key not in self._device_tables:
            self._device_tables[key] = (self.roles_to_verb_tensor_list.to(device=device, dtype=torch.int32).contiguous(),
                                        self.adj_table.to(device).contiguous(), self.role_counts.to(device))
        return self._device_tables[key]

    # ------------------------------------------------------------------ reference surface
    def get_max_role_count(self): return self.max_role_count          # imsitu_encoder.py:146
    def get_num_verbs(self): return len(self.verb_list)               # :149
    def get_num_roles(self): return len(self.role_list)               # :152
    def get_num_labels(self): return len(self.label_list)             # :155

    def get_role_count(self, verb_id):                                # :158
        return int(self.role_counts[int(verb_id)])

    def get_role_ids(self, verb_id):                                  # :168
        return self.roles_to_verb_tensor_list[verb_id]

    def get_role_ids_batch(self, verbs):                              # :172 (vectorised gather)
        verbs = torch.as_tensor(verbs)
        return self.roles_to_verb_tensor_list.to(verbs.device)[verbs.reshape(-1)]

    def get_adj_matrix_noself(self, verb_ids):                        # :209 (vectorised gather)
        verb_ids = torch.as_tensor(verb_ids)
        return self.adj_table.to(verb_ids.device)[verb_ids.reshape(-1)]

    def get_verb2role_encoding_batch(self, verb_ids):                 # :231
        return torch.stack([self.verb2role_encoding[int(v)] for v in verb_ids]).float()

    def get_label_ids(self, verb, frames):                            # :182-207
        roles, L = self.roles_per_verb[verb], len(self.label_list)
        out = torch.full((len(frames), self.max_role_count), L, dtype=torch.int64)
        for f, frame in enumerate(frames):
            for r, role in enumerate(roles):
                lab = frame[role]
                out[f, r] = self._label_id[lab] if lab in self._label_id else self._label_id['UNK']
        return out

    def encode(self, item):                                           # :161-166
        return self._verb_id[item['verb']], self.get_label_ids(item['verb'], item['frames'])

    # JSON round trip instead of pickling the object (reference sr.py:442-447 pickles it with torch.save)
    def state(self):
        return dict(verb_list=self.verb_list, role_list=self.role_list, label_list=self.label_list,
                    roles_per_verb=self.roles_per_verb)

    @classmethod
    def from_state(cls, st):
        e = cls()
        e.verb_list, e.role_list, e.label_list = list(st["verb_list"]), list(st["role_list"]), list(st["label_list"])
        e.roles_per_verb = {k: list(v) for k, v in st["roles_per_verb"].items()}
        e._finish()
        return e
