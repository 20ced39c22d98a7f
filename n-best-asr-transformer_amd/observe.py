"""Per-epoch observability files.

Mirrors /root/reference/utils/dataset/tod_asr_util.py:150-223: after every evaluation pass the reference dumps

* ``epoch_{e}_for_{split}_observe_{name}.csv`` — one row per utterance with the epoch metrics repeated on every row
  and the utterance's raw n-best text, predicted labels, gold labels and the exact-match flag;
* ``classification_report_epoch_{e}_for_{split}.txt`` — per-label precision / recall / F1 / support over the labels
  that occur in the gold annotations, as a ``tabulate`` table sorted by label.

Host-side file output only (nothing here touches the GPU).  pandas / tabulate are what the reference formats with, so
the byte layout of both files is theirs; the per-label counts are computed directly instead of through scikit-learn.
"""
import os


class EpochInfoCollector:
    """tod_asr_util.py:225-241 — plain record of one evaluation pass"""

    def __init__(self, raw_inputs, whole_pred_classes, true_golds, matches, mean_loss, precision, recall, f1, acc):
        self.raw_inputs = raw_inputs
        self.whole_pred_classes = whole_pred_classes
        self.true_golds = true_golds
        self.matches = matches
        self.mean_loss = mean_loss
        self.precision = precision
        self.recall = recall
        self.f1 = f1
        self.acc = acc

    @classmethod
    def from_cases(cls, cases, mean_loss, prf, acc):
        """``cases`` = [(raw_tokens, pred_labels, gold_labels)] as eval_epoch collects them"""
        return cls([" ".join(r) for r, _, _ in cases], [p for _, p, _ in cases], [g for _, _, g in cases],
                   [set(p) == set(g) for _, p, g in cases], mean_loss, prf[0], prf[1], prf[2], acc)


def label_metrics(golds, preds):
    """{label: (precision, recall, f1, support)} over labels seen in ``golds`` (tod_asr_util.py:150-197).

    A predicted label that never occurs in the gold set of the whole split is skipped, as in the reference; counts are
    on de-duplicated label sets per utterance.  Zero denominators give 0 (``zero_division=0``).
    """
    known = set(l for g in golds for l in g)
    tp = dict.fromkeys(known, 0)
    fp = dict.fromkeys(known, 0)
    fn = dict.fromkeys(known, 0)
    for g, p in zip(golds, preds):
        g, p = set(g), set(p)
        for l in g:
            if l in p:
                tp[l] += 1
            else:
                fn[l] += 1
        for l in (p - g) & known:
            fp[l] += 1
    out = {}
    for l in sorted(known):
        pr = tp[l] / (tp[l] + fp[l]) if tp[l] + fp[l] else 0.0
        rc = tp[l] / (tp[l] + fn[l]) if tp[l] + fn[l] else 0.0
        den = 2 * tp[l] + fn[l] + fp[l]                      # F1 from the confusion counts (scikit-learn's form)
        f1 = 2.0 * tp[l] / den if den else 0.0
        out[l] = (round(pr, 2), round(rc, 2), round(f1, 2), tp[l] + fn[l])
    return out


def classification_report(golds, preds):
    from tabulate import tabulate
    table = [[l, p, r, f, s] for l, (p, r, f, s) in label_metrics(golds, preds).items()]
    return tabulate(table, ["label", "precision", "recall", "f1-score", "support"])


def observability_lens(eic, epoch, dataset_type, output_dir, extra_name):
    """tod_asr_util.py:200-222: write the per-utterance CSV and the per-label report for one evaluation pass"""
    import pandas as pd
    n = len(eic.raw_inputs)
    df = pd.DataFrame({
        "epoch": [epoch] * n, "dataset": [dataset_type] * n, "mean_loss": [eic.mean_loss] * n,
        "precision": [eic.precision] * n, "recall": [eic.recall] * n, "f1": [eic.f1] * n, "acc": [eic.acc] * n,
        "raw_inputs": list(eic.raw_inputs), "pred_classes": list(eic.whole_pred_classes), "gold": list(eic.true_golds),
        "matches": list(eic.matches)})
    df.to_csv(os.path.join(output_dir, "epoch_%s_for_%s_observe_%s.csv" % (epoch, dataset_type, extra_name)), index=False)
    with open(os.path.join(output_dir, "classification_report_epoch_%s_for_%s.txt" % (epoch, dataset_type)), "w") as fp:
        fp.write(classification_report(eic.true_golds, eic.whole_pred_classes))
