#!/usr/bin/env python3
"""`Augmentation.py <image|dataset_dir> [-out DIR] [-seed N] [--workers N]` on the GPU.

Same flags, outputs and exit codes as the reference CLI (srcs/cli/Augmentation.py:32-200):
single-image mode writes original_<name> + six <transform>_<name> files; dataset mode writes
the balanced tree, artifacts/datasets/manifest_augmented.json and the distribution CSV.
"""
from __future__ import annotations

import argparse
import shutil
import sys
from pathlib import Path

from ..preprocessing.dataset_balancer import DatasetBalancer
from ..preprocessing.image_augmenter import ImageAugmenter
from ..utils.common import get_logger, setup_logging
from ..utils.distribution import count_images, merge_csv
from ..utils.ranks import init_from_env

logger = get_logger(__name__)

SUPPORTED_IMAGE_EXTENSIONS = {".jpg", ".jpeg", ".png", ".bmp", ".tiff"}
DEFAULT_DATASET_OUTPUT = "artifacts/augmented_directory"
DEFAULT_SINGLE_OUTPUT = "artifacts/example"
DEFAULT_SEED = 42
TRANSFORMATIONS = ["flip", "rotate", "skew", "shear", "crop", "distortion"]


class AugmentationError(Exception):
    pass


class InputValidationError(AugmentationError):
    pass


class ProcessingError(AugmentationError):
    pass


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Apply augmentations to balance a dataset (GPU).")
    parser.add_argument("input_path", help="Dataset root directory (PLANT/CLASS/*.jpg) or one image")
    parser.add_argument("-out", "--output", help="Output directory")
    parser.add_argument("-seed", "--seed", type=int, default=DEFAULT_SEED)
    parser.add_argument("--workers", type=int, default=None,
                        help="Host threads for JPEG decode/encode (default: auto)")
    return parser.parse_args(argv)


def single_image_mode(args, image_path: Path) -> None:
    output_dir = Path(args.output) if args.output else Path(DEFAULT_SINGLE_OUTPUT)
    output_dir.mkdir(parents=True, exist_ok=True)
    shutil.copy2(image_path, output_dir / f"original_{image_path.name}")
    augmenter = ImageAugmenter(seed=args.seed)
    for transform in TRANSFORMATIONS:
        output_path = output_dir / f"{transform}_{image_path.name}"
        if not getattr(augmenter, transform)(str(image_path), str(output_path)):
            raise ProcessingError(f"Failed to apply {transform} transformation")
        logger.info(f"{transform.capitalize()} applied: {output_path}")
    logger.info("Single image augmentation completed successfully")


def analyze_distribution(target_dir: Path) -> None:
    if not target_dir.exists():
        logger.warning("Target directory doesn't exist: %s", target_dir)
        return
    rows = count_images(target_dir, None)
    if not rows:
        logger.warning("No images found in target directory")
        return
    csv_path = Path("artifacts") / "distribution" / "balanced_distribution.csv"
    merge_csv(rows, csv_path)
    logger.info("Distribution CSV written: %s", csv_path.resolve())
    logger.info("Total balanced images: %d", sum(n for _, _, n in rows))


def dataset_mode_dir(args, source_dir: Path) -> None:
    target_dir = Path(args.output) if args.output else Path(DEFAULT_DATASET_OUTPUT)
    if not source_dir.exists():
        raise InputValidationError(f"Source directory not found: {source_dir}")
    # under `python -m torch.distributed.run --nproc-per-node N` the task list is cut into one
    # contiguous share per GPU (no exchange step); a plain run is one process, one GPU
    rk = init_from_env()
    DatasetBalancer(source_dir=str(source_dir), target_dir=str(target_dir), seed=args.seed,
                    workers=args.workers).run()
    logger.info("Dataset augmentation completed successfully")
    if rk.rank != 0:
        return
    try:
        analyze_distribution(target_dir)
    except Exception as e:  # noqa: BLE001
        logger.warning(f"Distribution analysis failed: {e}")


def main(argv=None) -> None:
    setup_logging()
    try:
        args = parse_args(argv)
        input_path = Path(args.input_path)
        if not input_path.exists():
            raise InputValidationError(f"Input path not found: {input_path}")
        if input_path.is_file() and input_path.suffix.lower() in SUPPORTED_IMAGE_EXTENSIONS:
            single_image_mode(args, input_path)
            return
        if input_path.is_dir():
            dataset_mode_dir(args, input_path)
            return
        raise InputValidationError("Unsupported input. Provide a dataset directory or an image file.")
    except InputValidationError as e:
        logger.error(f"Input validation error: {e}")
        sys.exit(1)
    except ProcessingError as e:
        logger.error(f"Processing error: {e}")
        sys.exit(1)
    except Exception as e:  # noqa: BLE001
        logger.error(f"Unexpected error: {e}")
        sys.exit(1)


if __name__ == "__main__":
    main()
