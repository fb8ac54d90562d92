"""Dataset balancing on the GPU with the reference's `DatasetBalancer` contract
(srcs/preprocessing/dataset_balancer.py:19-207).

The task list is built exactly as the reference does (same iteration order, same calls to
the global `random` stream: `random.choice(source_images)` then `random.randint(0, 1e6)` per
task), so names, sources and per-task seeds are identical.  Execution differs: instead of a
process pool running one PIL op per worker, tasks are processed in chunks — host threads
decode JPEGs, each task's parameters are drawn with its own seed exactly like
`ImageAugmenter(seed)` would, same-(op, size) groups run as ONE batched GPU launch, host
threads encode the results (quality 95).  Pixels are bit-identical to the reference's.
"""
from __future__ import annotations

import random
import shutil
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from .dataset_components import AugmentationPlanner, DistributionAnalyzer, ManifestGenerator
from .image_augmenter import ImageAugmenter, apply_batch, draw_params
from ..utils.common import get_logger
from ..utils.image_utils import ImageLoader
from ..utils.system_info import get_optimal_worker_count

logger = get_logger(__name__)

CHUNK = 256  # tasks per GPU round


class DatasetBalancer:
    def __init__(self, manifest_path=None, source_dir="images", target_dir="augmented_directory",
                 seed=42, workers=None):
        self.manifest_path = Path(manifest_path) if manifest_path else None
        self.source_dir = Path(source_dir)
        self.target_dir = Path(target_dir)
        self.transformer = ImageAugmenter(seed=seed)  # seeds the global RNGs like the reference
        self.workers = self._validate_workers(workers)
        self.analyzer = DistributionAnalyzer(self.source_dir)
        self.planner = None
        self.manifest_generator = None
        self.plan: Dict = {}
        self.tasks: List[dict] = []
        self.completed = 0
        self.failed = 0

    def _validate_workers(self, workers):
        max_workers = get_optimal_worker_count()
        if workers is None:
            workers = max(1, max_workers // 2)
        else:
            workers = max(1, int(workers))
            if workers > max_workers:
                logger.warning(f"Requested {workers} workers, but only {max_workers} CPUs available")
                workers = max_workers
        logger.info(f"Using {workers} host threads for JPEG decode/encode (max available: {max_workers})")
        return workers

    def analyze_distribution(self):
        counts = self.analyzer.analyze()
        self.analyzer.display_distribution()
        return counts

    def calculate_plan(self):
        self.planner = AugmentationPlanner(self.analyzer.counts)
        self.plan = self.planner.calculate_plan()
        return self.plan

    def _prepare_target_directory(self):
        logger.info(f"Preparing target directory: {self.target_dir}")
        if self.target_dir.exists():
            shutil.rmtree(self.target_dir)
        if not self.source_dir.exists():
            raise FileNotFoundError(f"Source directory not found: {self.source_dir}")
        shutil.copytree(self.source_dir, self.target_dir)

    def _get_images_by_class(self):
        images_by_class = defaultdict(list)
        for plant_dir in self.target_dir.iterdir():
            if plant_dir.is_dir():
                for class_dir in plant_dir.iterdir():
                    if class_dir.is_dir():
                        images = list(class_dir.glob("*.JPG")) + list(class_dir.glob("*.jpg"))
                        images_by_class[class_dir.name] = images
        return images_by_class

    def build_tasks(self, images_by_class) -> List[dict]:
        """dataset_balancer.py:105-129 — consumes the global `random` stream in the same order."""
        tasks = []
        for class_name, transforms in self.plan.items():
            if class_name not in images_by_class:
                logger.warning(f"No images found for class '{class_name}'")
                continue
            source_images = images_by_class[class_name]
            class_dir = source_images[0].parent
            for transform_name, count in transforms.items():
                for i in range(count):
                    source_img = random.choice(source_images)
                    new_name = source_img.stem + f"_aug_{transform_name}_{i + 1}" + source_img.suffix
                    tasks.append({"source_img": str(source_img),
                                  "output_path": str(class_dir / new_name),
                                  "transform_name": transform_name,
                                  "class_name": class_name,
                                  "seed": random.randint(0, 1000000)})
        return tasks

    # ------------------------------------------------------------------ GPU execution
    # A chunk goes through three stages: JPEG decode (host threads), parameter draws + HIP
    # kernels (this thread), JPEG encode (host threads).  The stages of consecutive chunks
    # overlap: chunk i+1 is being decoded and chunk i-1 encoded while chunk i is on the GPU.
    @staticmethod
    def _decode(task):
        try:
            return ImageLoader.load_as_array(task["source_img"])
        except Exception as e:  # noqa: BLE001 — the reference counts any failure
            logger.error(f"Failed to process {task['source_img']} - {e}")
            return None

    @staticmethod
    def _encode(item):
        arr, path = item
        try:
            ImageLoader.save_array(arr, path)
            return True
        except Exception as e:  # noqa: BLE001
            logger.error(f"Failed: {path} - {e}")
            return False

    @staticmethod
    def _draw_seeded(job):
        """One task's parameters from its own seeded generators (same values as seeding the
        process-global streams with that seed, which is what the reference's worker does)."""
        op, w, h, seed = job
        return draw_params(op, w, h, random.Random(seed), np.random.RandomState(seed))

    def _gpu_stage(self, chunk: List[dict], images: List[Optional[np.ndarray]], pool) -> List[tuple]:
        """Draw every task's parameters (a fresh seeded RNG per task, like the reference's
        `_process_single_transformation`; seeded tasks are independent and drawn on the host
        threads, seed 0 = "unseeded" continues the global streams in order), run the ops
        batched by (transform, size), return (pixels, output_path) pairs for the encoder."""
        import torch
        groups: Dict[tuple, List[int]] = defaultdict(list)
        params: List[dict] = [None] * len(chunk)  # type: ignore[list-item]
        drawing = {}
        for k, (task, img) in enumerate(zip(chunk, images)):
            if img is None:
                self.failed += 1
                continue
            h, w, _ = img.shape
            if task["seed"]:
                drawing[k] = pool.submit(self._draw_seeded, (task["transform_name"], w, h, task["seed"]))
            else:
                params[k] = draw_params(task["transform_name"], w, h)
            groups[(task["transform_name"], h, w)].append(k)
        for k, fut in drawing.items():
            params[k] = fut.result()
        out: List[tuple] = []
        for (op, _h, _w), ks in groups.items():
            try:
                x = torch.from_numpy(np.stack([images[k] for k in ks])).cuda()
                res = apply_batch(op, x, [params[k] for k in ks])
                if op != "rotate":   # same-sized results: one device->host copy for the group
                    host = torch.stack(res).cpu().numpy()
                    out.extend((host[j], chunk[k]["output_path"]) for j, k in enumerate(ks))
                else:
                    out.extend((o.cpu().numpy(), chunk[k]["output_path"]) for o, k in zip(res, ks))
            except Exception as e:  # noqa: BLE001
                logger.error(f"Failed batch {op}: {e}")
                self.failed += len(ks)
        return out

    def _finish_encodes(self, futures) -> None:
        for f in futures:
            if f.result():
                self.completed += 1
            else:
                self.failed += 1

    def execute_balancing(self):
        if not self.plan:
            logger.info("No augmentation plan - skipping execution")
            return
        self._prepare_target_directory()
        self.tasks = self.build_tasks(self._get_images_by_class())
        total = len(self.tasks)
        logger.info(f"Starting GPU augmentation: {total} images to generate")
        state = (random.getstate(), np.random.get_state())
        chunks = [self.tasks[b:b + CHUNK] for b in range(0, total, CHUNK)]
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            decoding = [pool.submit(self._decode, t) for t in chunks[0]] if chunks else []
            encoding: list = []
            for i, chunk in enumerate(chunks):
                images = [f.result() for f in decoding]
                decoding = [pool.submit(self._decode, t) for t in chunks[i + 1]] if i + 1 < len(chunks) else []
                results = self._gpu_stage(chunk, images, pool)
                self._finish_encodes(encoding)
                encoding = [pool.submit(self._encode, r) for r in results]
                done = self.completed + self.failed
                if done % 500 < CHUNK and done:
                    logger.info(f"Progress: {done}/{total} ({done / total * 100:.1f}%) - "
                                f"{self.completed} success, {self.failed} failed")
            self._finish_encodes(encoding)
        random.setstate(state[0])
        np.random.set_state(state[1])
        logger.info(f"Augmentation complete: {self.completed} images generated, {self.failed} failed")
        self._generate_augmented_manifest()

    def _generate_augmented_manifest(self):
        self.manifest_generator = ManifestGenerator(self.analyzer.original_manifest, self.source_dir,
                                                    self.target_dir, self.workers)
        manifest = self.manifest_generator.generate_augmented_manifest()
        out_dir = self.manifest_path.parent if self.manifest_path is not None else Path("artifacts/datasets")
        out_dir.mkdir(parents=True, exist_ok=True)
        self.manifest_generator.save_manifest(manifest, out_dir / "manifest_augmented.json")

    def run(self):
        logger.info("=== Dataset Balancing System ===")
        try:
            self.analyze_distribution()
            self.calculate_plan()
            self.execute_balancing()
            logger.info("=== Balancing Complete ===")
        except Exception as e:
            logger.error(f"Dataset balancing failed - {e}")
            raise
