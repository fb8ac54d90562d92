#!/usr/bin/env python3
"""`python -m leaffliction_amd.cli.train` — the reference's train CLI on MI355X.

Same flags, presets and artifacts as srcs/cli/train.py:30-117,450-473.  One process per GPU:
launch with `python -m torch.distributed.run --nproc-per-node N ...` for data-parallel
training (global batch = --batch-size, sharded across ranks; RCCL all-reduce of the flat
gradient bucket); a plain `python -m ...` run is single-GPU.  Like the reference, training is
mixed-precision unless `--no-mixed-precision` is given: bf16 storage and MFMA operands, fp32
accumulation / variables / statistics (fp32 throughout when the shape cannot take the bf16 path).
Logs and returns 0 on FileNotFoundError/ValueError like the reference (train.py:471-473).
"""
from __future__ import annotations

import argparse
import logging
import os
import random
from pathlib import Path
from typing import Any, Dict, List, Tuple

import numpy as np

from ..dataio.manifest import build_label_mapping, load_manifest, select_items
from ..dataio.sequence import ManifestSequence
from ..model.cnn import adapt_normalization, build_leafcnn
from ..train.parallel import DataParallel
from ..train.utils import (CosineDecay, StopOnValAcc, build_callbacks, build_loss, build_optimizer,
                           save_best_variant)
from ..utils.common import setup_logging
from ..utils.system_info import get_optimal_worker_count

LOGGER = logging.getLogger(__name__)

REGULARIZED_CFG = {"optimizer": "adamw", "lr": 0.002, "weight_decay": 0.0001,
                   "label_smoothing": 0.02, "cosine_decay": True, "ema_decay": 0.999,
                   "clipnorm": 0.5, "cache": False}
FAST_OVERRIDE = {"optimizer": "adam", "lr": 3e-3, "weight_decay": 0.0, "label_smoothing": 0.0,
                 "cosine_decay": True, "ema_decay": 0.0, "clipnorm": 0.0, "cache": True}


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Train leaf_cnn (MI355X) using manifest_split.json")
    p.add_argument("--manifest", type=Path, default=Path("artifacts/datasets/manifest_augmented.json"))
    p.add_argument("--epochs", type=int, default=20)
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--img-size", type=int, default=224)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--no-normalization", action="store_true")
    p.add_argument("--no-mixed-precision", action="store_true")
    p.add_argument("--fast", action="store_true")
    p.add_argument("--scale", choices=["tiny", "small", "base"], default="base")
    mx = p.add_mutually_exclusive_group()
    mx.add_argument("--tiny", action="store_true")
    mx.add_argument("--small", action="store_true")
    mx.add_argument("--base", action="store_true")
    p.add_argument("--separable", action="store_true")
    p.add_argument("--target-val-acc", type=float, default=None)
    args = p.parse_args(argv)
    for s in ("tiny", "small", "base"):
        if getattr(args, s, False):
            args.scale = s
    return args


def validate_manifest(args) -> Path:
    if not args.manifest.exists():
        if args.manifest.name == "manifest_augmented.json":
            fallback = args.manifest.with_name("manifest_split.json")
            if fallback.exists():
                LOGGER.warning("Augmented manifest not found, falling back to: %s", fallback)
                return fallback
        LOGGER.error("Manifest not found: %s", args.manifest)
        raise FileNotFoundError(f"Manifest not found: {args.manifest}")
    return args.manifest


def prepare_data(manifest_path: Path) -> Tuple[List, List, Dict]:
    items = load_manifest(manifest_path)
    train_items, val_items = select_items(items, "train"), select_items(items, "val")
    if not train_items or not val_items:
        LOGGER.error("Insufficient data (train=%d, val=%d)", len(train_items), len(val_items))
        raise ValueError("Insufficient training or validation data")
    label2idx = build_label_mapping(train_items)
    LOGGER.info("Classes: %d", len(label2idx))
    return train_items, val_items, label2idx


def get_training_config(fast_mode: bool) -> Dict:
    cfg = REGULARIZED_CFG.copy()
    if fast_mode:
        cfg.update(FAST_OVERRIDE)
    LOGGER.info("Mode: %s -> %s", "FAST" if fast_mode else "REGULARIZED", cfg)
    return cfg


def get_model_parameters(scale: str) -> Tuple[List[int], float, float]:
    if scale == "tiny":
        return [16, 32, 64], 0.10, 0.30
    if scale == "small":
        return [32, 64, 128], 0.15, 0.35
    return [32, 64, 128, 256], 0.15, 0.40


def create_data_sequences(train_items, val_items, label2idx, args, cfg, num_classes, dp):
    seq_workers = get_optimal_worker_count()
    common = dict(num_classes=num_classes, one_hot=cfg["label_smoothing"] > 0.0, workers=seq_workers,
                  rank=dp.rank, world=dp.world)
    train_seq = ManifestSequence(train_items, label2idx, args.img_size, args.batch_size, shuffle=True,
                                 seed=args.seed, cache=cfg["cache"], **common)
    val_seq = ManifestSequence(val_items, label2idx, args.img_size, args.batch_size, shuffle=False,
                               seed=args.seed, cache=True, **common)
    return train_seq, val_seq


def build_and_compile_model(args, cfg: Dict, num_classes: int, train_seq: Any, dp) -> Any:
    widths, drop_block, drop_top = get_model_parameters(args.scale)
    model, norm_layer = build_leafcnn(num_classes=num_classes, img_size=args.img_size,
                                      use_norm=not args.no_normalization, widths=widths,
                                      drop_block=drop_block, drop_top=drop_top,
                                      l2_reg=cfg["weight_decay"], separable=args.separable,
                                      seed=args.seed)
    if norm_layer is not None:
        # every rank adapts on the same (unsharded) first batches so the statistics agree
        full = ManifestSequence(train_seq.items, train_seq.label2idx, args.img_size, args.batch_size,
                                shuffle=False, seed=args.seed, num_classes=num_classes,
                                one_hot=train_seq.one_hot, workers=train_seq.workers)
        full.indexes = list(train_seq.indexes)
        adapt_normalization(norm_layer, full)
    if not args.no_mixed_precision:
        # the reference's default policy (train.py:179-190: mixed_float16 unless --no-mixed-precision):
        # 16-bit storage and matrix operands, fp32 variables / statistics / loss — here in bf16
        try:
            model.set_training_dtype("bf16")
            LOGGER.info("Mixed precision: bf16 storage and MFMA operands, fp32 accumulation and variables")
        except ValueError as e:
            LOGGER.info("Mixed precision not available for this shape (%s): training in fp32", e)
    dp.broadcast_(model.flat_p, 0)
    model._mut += 1   # written by a collective: what inference keeps of the parameters is stale
    if dp.active:
        # same initial weights everywhere; independent dropout / in-model augmentation draws per shard
        model.reseed_step_rng(args.seed + dp.rank)
    steps_per_epoch = len(train_seq)
    base_lr = CosineDecay(cfg["lr"], steps_per_epoch * args.epochs) if cfg["cosine_decay"] else cfg["lr"]
    model.compile(optimizer=build_optimizer(cfg, base_lr), loss=build_loss(cfg), metrics=["accuracy"])
    return model


def create_training_metadata(args, cfg, num_classes, train_items, val_items, dp) -> Dict:
    widths, drop_block, drop_top = get_model_parameters(args.scale)
    return {
        "run": {"seed": args.seed, "epochs": args.epochs, "batch_size": args.batch_size},
        "data": {"manifest": str(args.manifest.resolve()), "img_size": args.img_size,
                 "num_classes": num_classes, "train_items": len(train_items),
                 "val_items": len(val_items)},
        "model": {"name": "leaf_cnn", "scale": args.scale, "separable": bool(args.separable),
                  "use_normalization": not args.no_normalization, "widths": widths,
                  "drop_block": drop_block, "drop_top": drop_top, "l2": cfg["weight_decay"]},
        "training": {"optimizer": cfg["optimizer"], "base_lr": cfg["lr"],
                     "cosine_decay": bool(cfg["cosine_decay"]),
                     "label_smoothing": cfg["label_smoothing"], "ema_decay": cfg["ema_decay"],
                     "clipnorm": cfg["clipnorm"], "mixed_precision": False},
        "system": {"sequence_workers": get_optimal_worker_count(), "backend": "hip/gfx950",
                   "world_size": dp.world},
    }


def main(argv=None) -> None:
    args = parse_args(argv)
    setup_logging()
    random.seed(args.seed)
    np.random.seed(args.seed)
    import torch
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LEAFFLICTION_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs
    # than ranks (ranks share the cards); production runs are one GPU per rank over RCCL
    backend = os.environ.get("LEAFFLICTION_DIST_BACKEND")
    dev_index = local_rank if backend in (None, "nccl") else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dp = DataParallel(backend=backend, device=torch.device("cuda", dev_index))
    try:
        manifest_path = validate_manifest(args)
        train_items, val_items, label2idx = prepare_data(manifest_path)
        num_classes = len(label2idx)
        cfg = get_training_config(args.fast)
        train_seq, val_seq = create_data_sequences(train_items, val_items, label2idx, args, cfg,
                                                   num_classes, dp)
        model = build_and_compile_model(args, cfg, num_classes, train_seq, dp)
        meta = create_training_metadata(args, cfg, num_classes, train_items, val_items, dp)
        meta["training"]["mixed_precision"] = model.train_dtype == "bf16"
        callbacks, ema_cb = build_callbacks(cfg)
        if getattr(args, "target_val_acc", None):
            callbacks.append(StopOnValAcc(args.target_val_acc))
        history = model.fit(train_seq, validation_data=val_seq, epochs=args.epochs, callbacks=callbacks,
                            dp=dp)
        save_best_variant(model, val_seq, ema_cb, out_dir=Path("artifacts/models"),
                          label2idx=label2idx, history=history, meta=meta, dp=dp)
    except (FileNotFoundError, ValueError) as e:
        LOGGER.error("Training failed: %s", e)
    finally:
        dp.shutdown()


if __name__ == "__main__":
    main()
