"""Classification metrics with the reference's keys (srcs/utils/metrics.py:37-93), computed
from integer confusion counts with numpy (same values as sklearn with zero_division=0; labels
present in y_true or y_pred define the averaged set, as sklearn does)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def compute_classification_metrics(y_true: List[int], y_pred: List[int],
                                   labels: List[str]) -> Dict[str, float]:
    yt, yp = np.asarray(y_true, dtype=np.int64), np.asarray(y_pred, dtype=np.int64)
    present = np.unique(np.concatenate([yt, yp])) if yt.size else np.array([], np.int64)
    n = int(max(len(labels), (present.max() + 1) if present.size else 0))
    cm = np.zeros((n, n), dtype=np.int64)
    np.add.at(cm, (yt, yp), 1)
    tp = np.diag(cm).astype(np.float64)
    pred_tot, true_tot = cm.sum(0).astype(np.float64), cm.sum(1).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = np.where(pred_tot > 0, tp / pred_tot, 0.0)
        rec = np.where(true_tot > 0, tp / true_tot, 0.0)
        f1 = np.where(prec + rec > 0, 2 * prec * rec / (prec + rec), 0.0)
    sel = present
    w = true_tot[sel]

    def macro(v):
        return float(v[sel].mean()) if sel.size else 0.0

    def weighted(v):
        return float((v[sel] * w).sum() / w.sum()) if w.sum() > 0 else 0.0

    metrics = {"accuracy": float((yt == yp).mean()) if yt.size else 0.0,
               "macro_f1": macro(f1), "weighted_f1": weighted(f1),
               "macro_precision": macro(prec), "weighted_precision": weighted(prec),
               "macro_recall": macro(rec), "weighted_recall": weighted(rec)}
    if len(labels) == 2 and n >= 2:
        metrics["binary_f1"] = float(f1[1])
        metrics["binary_precision"] = float(prec[1])
        metrics["binary_recall"] = float(rec[1])
    for j, i in enumerate(sel.tolist()):  # sklearn's average=None is indexed by present labels
        if j < len(labels):
            metrics[f"f1_{labels[j]}"] = float(f1[i])
            metrics[f"precision_{labels[j]}"] = float(prec[i])
            metrics[f"recall_{labels[j]}"] = float(rec[i])
    return metrics
