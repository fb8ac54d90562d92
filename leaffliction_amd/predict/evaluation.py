"""Evaluation against ground truth (mirror of srcs/predict/evaluation.py:14-144)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional

from ..utils.common import get_logger
from ..utils.metrics import compute_classification_metrics

logger = get_logger(__name__)


class PredictionEvaluator:
    def __init__(self, predictor):
        self.predictor = predictor

    def evaluate_predictions(self, image_paths: List[Path], true_labels: List[str],
                             output_dir: Optional[Path] = None) -> Dict[str, float]:
        if len(image_paths) != len(true_labels):
            raise ValueError("Number of images must match number of true labels")
        predictions = self.predictor.predict_batch(image_paths)
        labels = self.predictor.model_loader.labels
        label_to_idx = {lab: i for i, lab in enumerate(labels)}
        y_true, y_pred, valid = [], [], []
        for i, (t, pr) in enumerate(zip(true_labels, [p["top_prediction"] for p in predictions])):
            if t not in label_to_idx or pr not in label_to_idx:
                logger.warning(f"Skipping unknown label: {t} or {pr}")
                continue
            y_true.append(label_to_idx[t])
            y_pred.append(label_to_idx[pr])
            valid.append((i, predictions[i]))
        if not y_true:
            logger.error("No valid predictions to evaluate")
            return {}
        metrics = compute_classification_metrics(y_true, y_pred, labels)
        if output_dir:
            output_dir = Path(output_dir)
            output_dir.mkdir(parents=True, exist_ok=True)
            results = {"metrics": metrics,
                       "evaluation_info": {"total_images": len(image_paths),
                                           "valid_predictions": len(valid), "class_labels": labels},
                       "detailed_results": [
                           {"image_path": str(pred["image_path"]), "true_label": true_labels[i],
                            "predicted_label": pred["top_prediction"],
                            "confidence": pred["confidence"],
                            "correct": true_labels[i] == pred["top_prediction"]}
                           for i, pred in valid]}
            with (output_dir / "evaluation_results.json").open("w", encoding="utf-8") as f:
                json.dump(results, f, indent=2)
        return metrics


def evaluate_from_manifest(predictor, manifest_path: Path, split: str = "test",
                           output_dir: Optional[Path] = None) -> Dict[str, float]:
    with Path(manifest_path).open("r", encoding="utf-8") as f:
        data = json.load(f)
    items = data["items"] if isinstance(data, dict) and "items" in data else data
    sel = [it for it in items if it.get("split") == split]
    if not sel:
        logger.error(f"No items found for split '{split}' in manifest")
        return {}
    return PredictionEvaluator(predictor).evaluate_predictions(
        [Path(it["src"]) for it in sel], [it.get("label", it["class"]) for it in sel], output_dir)


def sharded_confusion_counts(predictor, image_paths: List[Path], true_labels: List[str], ranks=None):
    """Integer confusion counts `cm[true][pred]` over a file list cut into one contiguous share
    per GPU replica: every replica predicts its share, counts locally, and the counts are SUMMED
    over ranks (integers: exact, order-independent) — SURVEY §8e, BASELINE configs[4].  Returns
    the full matrix (list of lists) on every rank; unknown labels are skipped like
    `evaluate_predictions` does."""
    from ..utils import ranks as R
    rk = ranks or R.current()
    labels = predictor.model_loader.labels
    index = {lab: i for i, lab in enumerate(labels)}
    c = len(labels)
    b, e = R.contiguous_share(len(image_paths), rk.rank, rk.world)
    counts = [0] * (c * c)
    if e > b:
        preds = predictor.predict_batch([Path(p) for p in image_paths[b:e]])
        by_path = {str(r["image_path"]): r["top_prediction"] for r in preds}
        for p, t in zip(image_paths[b:e], true_labels[b:e]):
            pr = by_path.get(str(Path(p)))
            if pr in index and t in index:
                counts[index[t] * c + index[pr]] += 1
    total = rk.sum_ints(counts)
    return [total[i * c:(i + 1) * c] for i in range(c)]
