#!/usr/bin/env python3
"""`predict.py image_or_dir [-learnings DIR] [-out DIR] [-json P] [-batch] [--evaluate ...]`.

Flags, JSON schema (`batch_results` + `summary`), the sampled accuracy gate and the exit
codes (1 on error, 2 when the gate is never reached) follow srcs/cli/predict.py:17-87,
305-436,492-563.  Montage / dashboard rendering is presentation and not reproduced.
"""
from __future__ import annotations

import argparse
import json
import random as _random
import sys
import time
from pathlib import Path
from typing import Optional

from ..predict.evaluation import PredictionEvaluator
from ..predict.predictor import Predictor
from ..utils.common import get_logger, setup_logging
from ..utils.image_utils import ImageLoader
from ..utils.ranks import init_from_env

logger = get_logger(__name__)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Predict leaf disease from image(s) (MI355X)")
    p.add_argument("image_path")
    p.add_argument("-learnings", "--learnings-dir", default="artifacts/models")
    p.add_argument("-out", "--output-dir", default="artifacts/prediction_output")
    p.add_argument("-json", "--json-output", default="artifacts/prediction_output/batch_results.json")
    p.add_argument("-batch", "--batch-mode", action="store_true")
    p.add_argument("--evaluate", action="store_true")
    p.add_argument("--manifest")
    p.add_argument("--split", default="val")
    p.add_argument("--sample-size", type=int, default=100)
    p.add_argument("--target-acc", type=float, default=0.90)
    p.add_argument("--max-attempts", type=int, default=5)
    return p.parse_args(argv)


def validate_inputs(args):
    image_path, learnings_dir = Path(args.image_path), Path(args.learnings_dir)
    if not image_path.exists():
        raise FileNotFoundError(f"Path not found: {image_path}")
    if args.batch_mode and not image_path.is_dir():
        raise ValueError(f"Batch mode requires a directory, got: {image_path}")
    if not args.batch_mode and not image_path.is_file():
        raise ValueError(f"Single mode requires an image file, got: {image_path}")
    if not learnings_dir.exists():
        raise FileNotFoundError(f"Learnings directory not found: {learnings_dir}")
    if not (learnings_dir / "meta.json").exists():
        raise FileNotFoundError(f"Meta file not found: {learnings_dir / 'meta.json'}")
    if args.evaluate:
        if not args.batch_mode:
            raise ValueError("--evaluate requires --batch-mode")
        if not args.manifest:
            raise ValueError("--evaluate requires --manifest")
        if not Path(args.manifest).exists():
            raise FileNotFoundError(f"Manifest not found: {args.manifest}")
    return image_path, learnings_dir


def create_batch_summary(results, processing_time):
    if not results:
        return {"total_images": 0, "processing_time": f"{processing_time:.2f}s"}
    counts = {}
    for r in results:
        counts[r["top_prediction"]] = counts.get(r["top_prediction"], 0) + 1
    avg = sum(r["confidence"] for r in results) / len(results)
    return {"total_images": len(results), "processing_time": f"{processing_time:.2f}s",
            "average_confidence": f"{avg:.2%}", "prediction_distribution": counts}


def save_batch_results_json(results, processing_time, output_path):
    output_path = Path(output_path)
    if not output_path.is_absolute() and not str(output_path).startswith("artifacts/"):
        output_path = Path("artifacts/prediction_output") / output_path.name
    data = {"batch_results": [{"image_path": str(r["image_path"]),
                               "top_prediction": r["top_prediction"],
                               "confidence": r["confidence"],
                               "all_probabilities": r["all_probabilities"]} for r in results],
            "summary": create_batch_summary(results, processing_time)}
    output_path.parent.mkdir(parents=True, exist_ok=True)
    with open(output_path, "w") as f:
        json.dump(data, f, indent=2)
    return output_path


def _item_path(item, image_dir: Path, manifest_path: Optional[Path]) -> Optional[Path]:
    raw = next((item[k] for k in ("src", "id", "path", "filepath", "file", "image", "img_path")
                if k in item), None)
    if not raw:
        return None
    p = Path(raw)
    if p.is_absolute():
        return p if p.exists() else None
    for base in ([manifest_path.parent] if manifest_path else []) + [image_dir]:
        if (base / p).exists():
            return base / p
    return p if p.exists() else None


def _load_manifest_items(manifest_path, split):
    with open(manifest_path, "r") as f:
        data = json.load(f)
    raw = data["items"] if isinstance(data, dict) and "items" in data else (
        data if isinstance(data, list) else [])
    if split is None:
        return list(raw)
    items = [it for it in raw if it.get("split") == split]
    return items if items else list(raw)


def run_sampling_enforced_batch(predictor, image_dir: Path, manifest_path: Path, split: str,
                                sample_size: int, target_acc: float, max_attempts: int,
                                json_output: Optional[str], ranks=None) -> bool:
    """predict.py:305-388: sample, predict, emit outputs only once accuracy >= target.  With
    replicas (`ranks`), rank 0's wall-clock sample seed is shared, every replica predicts its
    contiguous share of the sample, and rank 0 alone writes the outputs."""
    from ..utils import ranks as R
    rk = ranks or R.Solo()
    best = 0.0
    for attempt in range(1, int(max_attempts) + 1):
        logger.info("Sampling attempt %d/%d (n=%d)", attempt, int(max_attempts), int(sample_size))
        items = _load_manifest_items(manifest_path, split)
        rng = _random.Random(rk.broadcast_object(int(time.time()) % 1_000_000))
        sampled = rng.sample(items, min(int(sample_size), len(items))) if items else []
        paths, labels = [], []
        for it in sampled:
            p = _item_path(it, image_dir, manifest_path)
            if p is not None and p.exists():
                paths.append(p)
                labels.append(it.get("label", it.get("class")))
        if not paths:
            logger.warning("Sampling produced no valid images; retrying...")
            continue
        t0 = time.time()
        results = predictor.predict_batch_sharded(paths, rk)
        proc = time.time() - t0
        acc = sum(r.get("top_prediction") == t for r, t in zip(results, labels)) / max(len(results), 1)
        logger.info("Sample accuracy: %.4f on %d images", acc, len(results))
        if acc >= float(target_acc):
            if rk.rank != 0:
                return True
            if json_output:
                logger.info("Results saved to: %s", save_batch_results_json(results, proc, json_output))
            try:
                PredictionEvaluator(predictor).evaluate_predictions(
                    paths, labels, output_dir=Path("artifacts/prediction_output/evaluation"))
            except Exception as e:  # noqa: BLE001
                logger.warning("Detailed evaluation failed: %s", e)
            logger.info("Batch prediction completed successfully")
            return True
        best = max(best, acc)
    logger.error("Failed to reach target accuracy %.2f after %d attempts (best=%.4f). "
                 "No outputs emitted.", float(target_acc), int(max_attempts), float(best))
    return False


def main(argv=None) -> None:
    setup_logging()
    try:
        args = parse_args(argv)
        image_path, learnings_dir = validate_inputs(args)
        rk = init_from_env()   # replicas under torch.distributed.run; one process otherwise
        predictor = Predictor(learnings_dir)
        predictor.load()
        if args.batch_mode:
            if args.evaluate:
                ok = run_sampling_enforced_batch(predictor, image_path, Path(args.manifest), args.split,
                                                 args.sample_size, args.target_acc,
                                                 args.max_attempts, args.json_output, ranks=rk)
                if not ok:
                    sys.exit(2)
                return
            files = ImageLoader.get_image_files(image_path)
            if not files:
                logger.warning(f"No image files found in {image_path}")
                return
            t0 = time.time()
            results = predictor.predict_batch_sharded(files, rk)
            proc = time.time() - t0
            if rk.rank != 0:
                return
            out = save_batch_results_json(results, proc, args.json_output)
            logger.info("Results saved to: %s", out)
            for k, v in create_batch_summary(results, proc).items():
                logger.info("  %s: %s", k, v)
        else:
            r = predictor.predict_single(image_path)
            logger.info(f"Image: {r['image_path']}")
            logger.info(f"Prediction: {r['top_prediction']} ({r['confidence']:.2%})")
            for name, prob in sorted(r["all_probabilities"].items(), key=lambda x: -x[1])[:3]:
                logger.info(f"    {name}: {prob:.2%}")
    except SystemExit:
        raise
    except (FileNotFoundError, ValueError) as e:
        logger.error(f"Error: {e}")
        sys.exit(1)
    except Exception as e:  # noqa: BLE001
        logger.error(f"Unexpected error: {e}")
        sys.exit(1)


if __name__ == "__main__":
    main()
