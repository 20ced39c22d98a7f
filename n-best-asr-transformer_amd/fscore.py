"""Micro-averaged tuple F-score, the evaluation of /root/reference/utils/fscore.py:2-21
(predicted vs gold ``act[-slot[-value]]`` label lists; percentages; all zero when TP == 0)."""


def update_f1(pred, gold, TP, FP, FN):
    hits = sum(1 for term in pred if term in gold)
    TP += hits
    FP += len(pred) - hits
    FN += sum(1 for term in gold if term not in pred)
    return TP, FP, FN


def compute_f1(TP, FP, FN):
    if TP == 0:
        return 0, 0, 0
    return 100 * TP / (TP + FP), 100 * TP / (TP + FN), 100 * 2 * TP / (2 * TP + FN + FP)
