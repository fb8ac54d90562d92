"""Training utilities with the reference's names (srcs/train/utils.py:17-130).

`build_optimizer` / `build_loss` return plain config dicts consumed by `LeafCNN.compile`
(the optimizer itself is the fused AdamW+clipnorm+EMA kernel); callbacks follow the Keras
callback protocol the reference relies on; `save_best_variant` writes the same artifact set:
leaf_cnn.keras, labels.json, history.json, meta.json, confusion_matrix.json.
"""
from __future__ import annotations

import json
import logging
import math
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

from ..utils.confusion_matrix import (_gather_predictions_and_labels, compute_confusion_counts,
                                      plot_confusion_png, save_confusion_json)

LOGGER = logging.getLogger(__name__)


class CosineDecay:
    """keras.optimizers.schedules.CosineDecay(initial_lr, decay_steps, alpha=0) (train.py:313-318)."""

    def __init__(self, initial_learning_rate: float, decay_steps: int) -> None:
        self.initial_learning_rate = float(initial_learning_rate)
        self.decay_steps = max(1, int(decay_steps))

    def __call__(self, step: int) -> float:
        s = min(int(step), self.decay_steps)
        return self.initial_learning_rate * 0.5 * (1.0 + math.cos(math.pi * s / self.decay_steps))


def build_optimizer(cfg: Dict, base_lr) -> Dict[str, Any]:
    """AdamW(lr, weight_decay, clipnorm) or Adam(lr, clipnorm) as a config dict (utils.py:17-27)."""
    opt: Dict[str, Any] = {"name": "adamw" if cfg.get("optimizer") == "adamw" else "adam",
                           "clipnorm": cfg["clipnorm"] if cfg.get("clipnorm", 0.0) > 0 else 0.0,
                           "weight_decay": cfg.get("weight_decay", 0.0)
                           if cfg.get("optimizer") == "adamw" else 0.0,
                           "ema_decay": float(cfg.get("ema_decay", 0.0) or 0.0)}
    if callable(base_lr):
        opt["schedule"] = base_lr
    else:
        opt["lr"] = float(base_lr)
    return opt


def build_loss(cfg: Dict) -> Dict[str, Any]:
    """CategoricalCrossentropy(label_smoothing) or sparse CCE (utils.py:30-35)."""
    ls = float(cfg.get("label_smoothing", 0.0) or 0.0)
    return {"name": "categorical_crossentropy" if ls > 0 else "sparse_categorical_crossentropy",
            "label_smoothing": ls}


class Callback:
    model: Any = None

    def set_model(self, model) -> None:
        self.model = model

    def on_train_begin(self) -> None: ...
    def on_train_end(self) -> None: ...
    def on_train_batch_end(self, batch, logs=None) -> None: ...
    def on_epoch_end(self, epoch, logs=None) -> None: ...


class EMACallback(Callback):
    """utils.py:38-57.  The reference averages `model.get_weights()` on the host every batch;
    here the average lives on the device (updated inside the optimizer kernel) and
    `ema_weights` fetches it on demand."""

    def __init__(self, decay: float) -> None:
        self.decay = float(decay)

    @property
    def ema_weights(self) -> Optional[List]:
        m = self.model
        if self.decay <= 0.0 or m is None or not m.ema_started:
            return None
        return m.ema_weights()


class ReduceLROnPlateau(Callback):
    """keras ReduceLROnPlateau(patience=3, factor=0.3) monitoring val_loss.  With a learning-rate
    schedule Keras cannot set the LR: it is a warning + no-op here."""

    def __init__(self, patience: int = 3, factor: float = 0.3) -> None:
        self.patience, self.factor = patience, factor
        self.best, self.wait = math.inf, 0

    def on_epoch_end(self, epoch, logs=None) -> None:
        cur = (logs or {}).get("val_loss")
        if cur is None:
            return
        if cur < self.best - 1e-4:
            self.best, self.wait = cur, 0
            return
        self.wait += 1
        if self.wait >= self.patience:
            self.wait = 0
            opt = self.model._compiled.get("optimizer", {})
            if "schedule" in opt:
                LOGGER.warning("ReduceLROnPlateau: optimizer uses a LearningRateSchedule; "
                               "learning rate left unchanged")
            else:
                opt["lr"] = opt.get("lr", 1e-3) * self.factor
                LOGGER.info("ReduceLROnPlateau: lr -> %.3g", opt["lr"])


class EarlyStopping(Callback):
    """keras.callbacks.EarlyStopping(monitor="val_loss", patience=6, restore_best_weights=True)
    with Keras 3's bookkeeping (the reference pins keras>=3; utils.py:65): the first monitored
    epoch seeds `best_weights`; `wait` counts every epoch and is reset by an improvement; the
    stop is only raised after epoch 0; and the best weights are put back in `on_train_end`
    whenever they exist — also when training ran through all its epochs without stopping, so
    the "base" variant `save_best_variant` evaluates is the best-val_loss one, as in Keras."""

    def __init__(self, patience: int = 6, restore_best_weights: bool = True) -> None:
        self.patience, self.restore = patience, restore_best_weights
        self.on_train_begin()

    def on_train_begin(self) -> None:
        self.best, self.wait, self.best_weights = math.inf, 0, None
        self.best_epoch, self.stopped_epoch = 0, 0

    def _snapshot(self):
        return (self.model.flat_p.clone(), self.model.flat_s.clone())

    def on_epoch_end(self, epoch, logs=None) -> None:
        cur = (logs or {}).get("val_loss")
        if cur is None:
            return
        if self.restore and self.best_weights is None:
            self.best_weights, self.best_epoch = self._snapshot(), epoch
        self.wait += 1
        if cur < self.best:
            self.best, self.best_epoch, self.wait = cur, epoch, 0
            if self.restore:
                self.best_weights = self._snapshot()
            return
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.model.stop_training = True

    def on_train_end(self) -> None:
        if self.stopped_epoch > 0:
            LOGGER.info("EarlyStopping: stopped at epoch %d", self.stopped_epoch + 1)
        if self.restore and self.best_weights is not None:
            LOGGER.info("EarlyStopping: restoring the weights of epoch %d (best val_loss)",
                        self.best_epoch + 1)
            self.model.flat_p.copy_(self.best_weights[0])
            self.model.flat_s.copy_(self.best_weights[1])


class StopOnValAcc(Callback):
    """train.py:412-430 (--target-val-acc)."""

    def __init__(self, threshold: float) -> None:
        self.threshold = float(threshold)

    def on_epoch_end(self, epoch, logs=None) -> None:
        va = (logs or {}).get("val_accuracy")
        if va is not None and float(va) >= self.threshold:
            LOGGER.info("Target val_accuracy reached: %.4f >= %.4f; stopping", float(va), self.threshold)
            self.model.stop_training = True


def build_callbacks(cfg: Dict) -> Tuple[List[Callback], Optional[EMACallback]]:
    callbacks: List[Callback] = [ReduceLROnPlateau(patience=3, factor=0.3),
                                 EarlyStopping(patience=6, restore_best_weights=True)]
    ema_cb: Optional[EMACallback] = None
    decay = float(cfg.get("ema_decay", 0.0) or 0.0)
    if decay > 0.0:
        ema_cb = EMACallback(decay)
        callbacks.append(ema_cb)
    return callbacks, ema_cb


def save_best_variant(model, val_data, ema_cb: Optional[EMACallback], out_dir: Path,
                      label2idx: Dict[str, int], history, meta: Optional[Dict[str, Any]] = None,
                      dp=None) -> str:
    """Evaluate base vs EMA weights (EMA only if STRICTLY better), save the model and the JSON
    artifacts (utils.py:75-130).  Returns the saved variant."""
    best_acc = model.evaluate(val_data, dp=dp)[1]
    saved_variant = "base"
    ema_w = ema_cb.ema_weights if ema_cb is not None else None
    if ema_w is not None:
        original = model.get_weights()
        model.set_weights(ema_w)
        ema_acc = model.evaluate(val_data, dp=dp)[1]
        if float(ema_acc) <= float(best_acc):
            model.set_weights(original)
        else:
            saved_variant = "ema"
    # confusion counts over the (possibly rank-sharded) validation set: integer all-reduce
    labels_sorted = sorted(label2idx, key=lambda k: label2idx[k])
    y_true, y_pred = _gather_predictions_and_labels(model, val_data)
    cm = compute_confusion_counts(y_true, y_pred, num_classes=len(labels_sorted))
    if dp is not None and dp.active:
        import torch
        t = torch.tensor(cm, dtype=torch.int64, device=model.device)
        cm = dp.allreduce_counts(t).cpu().tolist()
    if dp is not None and dp.rank != 0:
        return saved_variant
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    model_path = out_dir / "leaf_cnn.keras"
    model.save(model_path)
    LOGGER.info("Model saved: %s", model_path.resolve())
    with (out_dir / "labels.json").open("w", encoding="utf-8") as f:
        json.dump({"label2idx": label2idx}, f, indent=2)
    with (out_dir / "history.json").open("w", encoding="utf-8") as f:
        json.dump({k: [float(x) for x in v] for k, v in history.history.items()}, f, indent=2)
    try:
        import torch
        meta_out: Dict[str, Any] = {
            "created_at": datetime.now(tz=timezone.utc).isoformat(),
            "model_file": str(model_path),
            "labels_file": str(out_dir / "labels.json"),
            "history_file": str(out_dir / "history.json"),
            "confusion_matrix_file": str(out_dir / "confusion_matrix.json"),
            "keras_version": "n/a (leaffliction_amd 0.1.0)",
            "tensorflow_version": f"n/a (torch {torch.__version__}, HIP)",
            "saved_variant": saved_variant,
            "labels": labels_sorted,
        }
        if meta:
            meta_out.update(meta)
        with (out_dir / "meta.json").open("w", encoding="utf-8") as f:
            json.dump(meta_out, f, indent=2)
    except (OSError, TypeError) as e:
        LOGGER.warning("Failed to write meta.json: %s", e)
    save_confusion_json(cm, labels_sorted, out_dir / "confusion_matrix.json")
    plot_confusion_png(cm, labels_sorted, out_dir / "confusion_matrix.png")
    return saved_variant
