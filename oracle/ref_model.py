"""CPU fp32 restatement of the reference hot path (reference `model.py`).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned to the reference by
tests/test_oracle_golden.py against fixtures written by oracle/gen_golden.py.

Knobs the reference hard-codes are arguments here (steps T: model.py:60 fixes 4;
backbone depth: model.py:16 fixes resnet152) so the larger BASELINE configs have
an oracle; at T=4 the functions reproduce the reference bit-for-bit up to fp32
summation order.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_resnet import RefResNet


class RefBackbone(nn.Module):
    """model.py:8-35 -- frozen ResNet whose `fc` is replaced by Identity, so the
    output is the pooled feature vector."""

    def __init__(self, depth=152, width=64, blocks=None):
        super().__init__()
        self.model = RefResNet(depth, width, blocks)
        for p in self.model.parameters():          # model.py:17-18
            p.requires_grad = False
        self.out_features = self.model.fc.in_features
        self.model.fc = nn.Identity()              # model.py:31

    def forward(self, x):                          # model.py:34-35
        return self.model(x)


class RefGGSNN(nn.Module):
    """model.py:38-86 -- gated graph network over the role graph."""

    NAMES = ("W_p", "W_z", "U_z", "W_r", "U_r", "W_h", "U_h")

    def __init__(self, layersize, steps=4):
        super().__init__()
        for n in self.NAMES:                       # model.py:47-56
            setattr(self, n, nn.Linear(layersize, layersize))
        self.steps = steps

    def neighbours(self, h, mask, verb):
        if verb:                                   # model.py:62-64
            return self.W_p(h)
        # model.py:66-77: for every target role i, project every (masked) source
        # role j separately (bias included each time) and sum over j.
        B, R = mask.shape[0], mask.shape[1]
        src = h.reshape(B, 1, R, -1) * mask.reshape(B, R, R, 1)   # [B, i, j, D]
        return self.W_p(src).sum(dim=2).reshape(B * R, -1)

    def neighbours_algebraic(self, h, mask):
        """Same quantity as `neighbours(..., verb=False)` by linearity:
        W_p(sum_j A_ij h_j) + R * b_p.  (What the HIP path computes.)"""
        B, R = mask.shape[0], mask.shape[1]
        agg = torch.bmm(mask, h.reshape(B, R, -1)).reshape(B * R, -1)
        return F.linear(agg, self.W_p.weight) + R * self.W_p.bias

    def forward(self, h, mask=None, verb=False):
        for _ in range(self.steps):                # model.py:60
            n = self.neighbours(h, mask, verb)
            z = torch.sigmoid(self.W_z(n) + self.U_z(h))          # model.py:80
            r = torch.sigmoid(self.W_r(n) + self.U_r(h))          # model.py:81
            c = torch.tanh(self.W_h(n) + self.U_h(r * h))         # model.py:82-83
            h = (1 - z) * h + z * c                               # model.py:84
        return h


class RefFCGGNN(nn.Module):
    """model.py:89-201.  `backbone_factory()` must return a module with
    `.out_features == D_hidden_state` mapping [B,3,H,W] -> [B,D]."""

    def __init__(self, encoder, D_hidden_state, steps=4, backbone_factory=None):
        super().__init__()
        self.encoder = encoder
        nr, nv, nl = encoder.get_num_roles(), encoder.get_num_verbs(), encoder.get_num_labels()
        self.role_emb = nn.Embedding(nr + 1, D_hidden_state, padding_idx=nr)   # model.py:95-97
        self.verb_emb = nn.Embedding(nv, D_hidden_state)                         # model.py:98
        backbone_factory = backbone_factory or (lambda: RefBackbone(152))
        self.convnet_verbs = backbone_factory()                                  # model.py:100
        self.convnet_nouns = backbone_factory()                                  # model.py:101
        self.ggsnn = RefGGSNN(D_hidden_state, steps)                             # model.py:103
        self.verb_classifier = nn.Sequential(nn.Dropout(0.5), nn.Linear(D_hidden_state, nv))    # 105-107
        self.nouns_classifier = nn.Sequential(nn.Dropout(0.5), nn.Linear(D_hidden_state, nl))   # 109-111

    def node_init(self, feat, verbs):
        """model.py:117-144: node[b,r] = relu(feat[b] * role_emb[role(b,r)] * verb_emb[verb_b])."""
        R = self.encoder.get_max_role_count()
        role_idx = self.encoder.get_role_ids_batch(verbs)                        # [B,R]
        v = self.verb_emb(verbs)                                                 # [B,D]
        ro = self.role_emb(role_idx)                                             # [B,R,D]
        node = F.relu(feat[:, None, :] * ro * v[:, None, :])
        return node.reshape(feat.shape[0] * R, -1)

    def predict_nouns(self, img, gt_verb, batch_size):                           # model.py:115-155
        feat = self.convnet_nouns(img)
        node = self.node_init(feat, gt_verb)
        mask = self.encoder.get_adj_matrix_noself(gt_verb)
        out = self.ggsnn(node, mask=mask, verb=False)
        logits = self.nouns_classifier(out)
        return logits.reshape(batch_size, self.encoder.get_max_role_count(), -1)

    def predict_verb(self, img, batch_size):                                     # model.py:158-168
        node = F.relu(self.convnet_verbs(img)).reshape(batch_size, -1)
        return self.verb_classifier(self.ggsnn(node, mask=None, verb=True))

    def forward(self, img, gt_verb):                                             # model.py:172-180
        B = img.shape[0]
        pred_verb = self.predict_verb(img, B)
        pred_nouns = self.predict_nouns(img, torch.argmax(pred_verb, 1), B)
        gt_pred_nouns = self.predict_nouns(img, gt_verb, B)
        return pred_verb, pred_nouns, gt_pred_nouns

    def verb_loss(self, pred_verb, gt_verb):                                     # model.py:183-187
        return F.cross_entropy(pred_verb, gt_verb)

    def nouns_loss(self, pred_nouns, gt_nouns):                                  # model.py:190-201
        L = self.encoder.get_num_labels()
        logits = pred_nouns.transpose(1, 2)                                      # [B,L,R]
        total = 0
        for a in range(3):                                                       # three annotators
            total = total + F.cross_entropy(logits, gt_nouns[:, a], ignore_index=L)
        return total


def train_step(model, optimizer, img, verb, nouns, max_norm=1.0):
    """sr.py:63-83 on CPU (autocast/GradScaler disable themselves there):
    zero_grad, forward, loss = verb_loss + nouns_loss (gt loss is only logged,
    sr.py:70,76), backward, clip_grad_norm_(model.parameters(), 1), step."""
    optimizer.zero_grad()
    pv, pn, pg = model(img, verb)
    vl, nl, gl = model.verb_loss(pv, verb), model.nouns_loss(pn, nouns), model.nouns_loss(pg, nouns)
    (vl + nl).backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    optimizer.step()
    return dict(verb_loss=vl.detach(), nouns_loss=nl.detach(), gt_nouns_loss=gl.detach(),
                grad_norm=gn.detach(), pred_verb=pv.detach(), pred_nouns=pn.detach(),
                gt_pred_nouns=pg.detach())
