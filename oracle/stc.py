"""Oracle: semantic-tuple-classifier (STC) heads, losses, label utilities and decoding.

Restates, in plain PyTorch fp32 on CPU:
  * HierarchicalClassifier   /root/reference/models/modules/hierarchical_classifier.py:6-60
  * convert_labels / reverse_top2bottom / onehot_to_scalar   /root/reference/utils/STC_util.py:4-51
  * cal_ce_loss / cal_total_loss   /root/reference/n_best_asr_bert.py:145-195 (loss objects :572-574)
  * pred_one_sample   /root/reference/n_best_asr_bert.py:198-215
  * update_f1 / compute_f1   /root/reference/utils/fscore.py:2-21

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class OracleHeads(nn.Module):
    """top sigmoid head + one softmax head per multi-value top label; final = top * softmax.
    Each linear sees its OWN dropout mask of the features (hierarchical_classifier.py:41,46)."""

    def __init__(self, top2bottom, in_dim, n_bottom, dropout):
        super().__init__()
        self.top2bottom = {int(k): list(v) for k, v in top2bottom.items()}
        self.n_bottom = n_bottom
        self.n_top = len(self.top2bottom)
        self.top_linear_layer = nn.Linear(in_dim, self.n_top)
        self.linear_layers = nn.ModuleDict(
            {"lin_%d" % k: nn.Linear(in_dim, len(v)) for k, v in self.top2bottom.items() if len(v) >= 2})
        self.p = dropout

    def forward(self, feats):
        drop = lambda t: F.dropout(t, self.p, self.training)
        top = torch.sigmoid(self.top_linear_layer(drop(feats)))
        bottoms = {k: torch.softmax(lin(drop(feats)), dim=1) for k, lin in self.linear_layers.items()}
        final = feats.new_empty(feats.size(0), self.n_bottom)
        for t in range(self.n_top):
            ids = self.top2bottom[t]
            if len(ids) >= 2:
                final[:, ids] = top[:, t:t + 1] * bottoms["lin_%d" % t]
            else:
                final[:, ids] = top[:, t:t + 1]
        return top, bottoms, final


def bottom2top_matrix(top2bottom):
    """STC_util.py:10-26 - 0/1 matrix [n_bottom, n_top]; a bottom label belongs to exactly one top."""
    b2t = {}
    for t, bs in top2bottom.items():
        for b in bs:
            if b in b2t:
                raise ValueError("map from bottom to top should be unique")
            b2t[b] = int(t)
    mat = torch.zeros(len(b2t), len(top2bottom))
    for b in range(len(b2t)):
        mat[b, b2t[b]] = 1
    return mat


def top_labels(bottom_labels, b2t):
    return bottom_labels @ b2t.to(bottom_labels.device)            # STC_util.py:4-7


def class_index(onehot):
    """STC_util.py:29-51 - index of the hot column; the LAST column (NONE) when the row is empty."""
    assert onehot.sum(dim=1).le(1).all()
    idx = onehot.max(dim=1)[1]
    return idx.masked_fill(onehot.sum(dim=1).eq(0), onehot.size(1) - 1)


def total_loss(top, bottoms, final, labels, top2bottom, b2t, asr_cls=None, trans_cls=None, add_l2_loss=False):
    """n_best_asr_bert.py:160-195.  Returns (loss_record: float, total: tensor, parts: dict).
    BCE sums (:572), NLL sum averaged over the heads (:157,:573), MSE mean (:574); NOT divided by B."""
    B = top.size(0)
    parts = {}
    total = 0.0
    if add_l2_loss and asr_cls is not None and trans_cls is not None:
        parts["mse"] = F.mse_loss(asr_cls, trans_cls)
        total = total + parts["mse"]
    parts["bottom_bce"] = F.binary_cross_entropy(final, labels, reduction="sum")
    total = total + parts["bottom_bce"]
    parts["top_bce"] = F.binary_cross_entropy(top, top_labels(labels, b2t), reduction="sum")
    total = total + parts["top_bce"]
    ces = []
    for t, ids in top2bottom.items():
        if len(ids) > 1:
            logp = torch.log(bottoms["lin_%s" % t] + 1e-12)
            ces.append(F.nll_loss(logp, class_index(labels[:, ids]), reduction="sum"))
    parts["ce"] = sum(ces) / len(ces)
    total = total + parts["ce"]
    record = sum(float(v.detach()) for v in parts.values()) / B
    return record, total, parts


def decode_indices(top, bottoms, top2bottom, idx2label):
    """Device-decodable form of pred_one_sample (:198-215): int [B, n_top], the predicted BOTTOM label
    index for every top label that fires (score > 0.5, strict), else -1.  Multi-value tops whose
    argmax label ends with 'NONE' give -1."""
    B, T = top.shape
    out = torch.full((B, T), -1, dtype=torch.long)
    for t in range(T):
        ids = top2bottom[t]
        fire = top[:, t] > 0.5
        if len(ids) == 1:
            out[:, t] = torch.where(fire, torch.tensor(ids[0]), torch.tensor(-1))
        else:
            am = bottoms["lin_%d" % t].argmax(dim=-1)
            real = torch.tensor(ids)[am]
            none = torch.tensor([idx2label[int(r)].endswith("NONE") for r in real])
            out[:, t] = torch.where(fire & ~none, real, torch.tensor(-1))
    return out


def pred_labels(i, top_row, bottoms, top2bottom, idx2label):
    """pred_one_sample (:198-215) verbatim semantics -> list[str] for sample i."""
    preds = []
    for t, p in enumerate(top_row):
        if p > 0.5:
            ids = top2bottom[t]
            if len(ids) == 1:
                preds.append(idx2label[ids[0]])
            else:
                lbl = idx2label[ids[int(bottoms["lin_%d" % t][i].argmax(dim=-1))]]
                if not lbl.endswith("NONE"):
                    preds.append(lbl)
    return preds


def update_f1(pred, gold, TP, FP, FN):        # utils/fscore.py:2-11
    for term in pred:
        if term in gold:
            TP += 1
        else:
            FP += 1
    for term in gold:
        if term not in pred:
            FN += 1
    return TP, FP, FN


def compute_f1(TP, FP, FN):                   # utils/fscore.py:14-21
    if TP == 0:
        return 0, 0, 0
    return 100 * TP / (TP + FP), 100 * TP / (TP + FN), 100 * 2 * TP / (2 * TP + FN + FP)
