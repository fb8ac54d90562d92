"""Confusion-matrix contract (mirror of srcs/utils/confusion_matrix.py:14-50,103-129).

Counts are integers [true][pred]; JSON = {"matrix": [[int]], "labels": [str]}.  The PNG is
presentation: drawn when matplotlib is installed, skipped otherwise.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List, Tuple

import numpy as np


def _gather_predictions_and_labels(model: Any, data: Any) -> Tuple[List[int], List[int]]:
    """Iterate (X, y) batches; y may be sparse (1-D) or one-hot (2-D)."""
    y_true: List[int] = []
    y_pred: List[int] = []
    for bx, by in data:
        probs = np.asarray(model.predict(bx))
        pred = np.argmax(probs, axis=-1)
        by = np.asarray(by)
        true = np.argmax(by, axis=-1) if by.ndim > 1 else by
        y_true.extend(int(v) for v in true.tolist())
        y_pred.extend(int(v) for v in pred.tolist())
    return y_true, y_pred


def compute_confusion_counts(y_true: List[int], y_pred: List[int],
                             num_classes: int) -> List[List[int]]:
    cm = [[0] * num_classes for _ in range(num_classes)]
    for t, p in zip(y_true, y_pred):
        cm[int(t)][int(p)] += 1
    return cm


def save_confusion_json(cm: List[List[int]], labels: List[str], out_path: Path) -> None:
    out_path = Path(out_path)
    out_path.parent.mkdir(parents=True, exist_ok=True)
    payload: Dict[str, Any] = {"matrix": cm, "labels": labels}
    with out_path.open("w", encoding="utf-8") as f:
        json.dump(payload, f, indent=2)


def confusion_matrix(model: Any, data: Any, labels: List[str], out_dir: Path) -> Path:
    out_dir = Path(out_dir)
    y_true, y_pred = _gather_predictions_and_labels(model, data)
    cm = compute_confusion_counts(y_true, y_pred, num_classes=len(labels))
    json_path = out_dir / "confusion_matrix.json"
    save_confusion_json(cm, labels, json_path)
    return json_path


def plot_confusion_png(cm: List[List[int]], labels: List[str], out_path: Path, *, normalize: bool = True) -> bool:
    """`confusion_matrix.png` next to the JSON (srcs/utils/confusion_matrix.py:51-97): rows
    normalised by default, cell values printed.  Presentation only — skipped (False) when
    matplotlib is not installed; the JSON is the contract."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception as exc:  # noqa: BLE001
        import logging
        logging.getLogger(__name__).warning("matplotlib unavailable, skipping confusion matrix PNG: %s", exc)
        return False
    k = len(labels)
    counts = np.asarray(cm, dtype=float)
    shown = counts / np.maximum(counts.sum(axis=1, keepdims=True), 1.0) if normalize else counts
    fig, ax = plt.subplots(figsize=(8, 6), dpi=150)
    img = ax.imshow(shown, cmap="Blues")
    plt.colorbar(img, ax=ax, fraction=0.046, pad=0.04)
    ax.set_xticks(range(k))
    ax.set_yticks(range(k))
    ax.set_xticklabels(labels, rotation=45, ha="right")
    ax.set_yticklabels(labels)
    ax.set_xlabel("Predicted")
    ax.set_ylabel("True")
    ax.set_title("Confusion Matrix" + (" (normalized)" if normalize else ""))
    for i in range(k):
        for j in range(k):
            ax.text(j, i, f"{shown[i, j]:.2f}" if normalize else f"{int(shown[i, j])}", ha="center",
                    va="center", color="black", fontsize=8)
    fig.tight_layout()
    Path(out_path).parent.mkdir(parents=True, exist_ok=True)
    fig.savefig(out_path)
    plt.close(fig)
    return True
