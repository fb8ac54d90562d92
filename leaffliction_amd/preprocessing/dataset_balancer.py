"""Dataset balancing on MI355X behind the reference's `DatasetBalancer` contract
(srcs/preprocessing/dataset_balancer.py:19-207): `DatasetBalancer(manifest_path, source_dir,
target_dir, seed, workers).run()` leaves the balanced tree and `manifest_augmented.json`.

What is kept from the reference, because outputs depend on it: the task list.  Plan order,
`random.choice(source_images)` then `random.randint(0, 10**6)` per task on the process-global
stream, and the `<stem>_aug_<transform>_<k><suffix>` names (dataset_balancer.py:105-129) —
so sources, names and per-task seeds are the reference's.

What is different: execution.  The reference hands one PIL call per task to a process pool
(:137-141).  Here the list is cut into one contiguous share per GPU (rank-0 builds it and
broadcasts it) and every share runs as a three-stage pipeline over chunks —
JPEG decode on host threads, one batched kernel launch per (transform, image size) group,
JPEG encode on host threads — with the stages of neighbouring chunks overlapping.  There is no
exchange step: ranks meet at a barrier, their success/failure counts are summed, and rank 0
writes the manifest.  Pixels are bit-identical to the reference's for every seeded task.
"""
from __future__ import annotations

import random
import shutil
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from .dataset_components import AugmentationPlanner, DistributionAnalyzer, ManifestGenerator
from .image_augmenter import ImageAugmenter, apply_batch, draw_params
from ..utils.common import get_logger
from ..utils.image_utils import ImageLoader
from ..utils import ranks as _ranks
from ..utils.ranks import contiguous_share
from ..utils.system_info import get_optimal_worker_count

logger = get_logger(__name__)

CHUNK = 256  # tasks per GPU round


class DatasetBalancer:
    def __init__(self, manifest_path=None, source_dir="images", target_dir="augmented_directory",
                 seed=42, workers=None):
        self.manifest_path = Path(manifest_path) if manifest_path else None
        self.source_dir, self.target_dir = Path(source_dir), Path(target_dir)
        # like the reference, constructing the augmenter is what seeds the global RNG streams
        # the task list is drawn from (image_augmenter.py:15-18)
        self.transformer = ImageAugmenter(seed=seed)
        self.workers = self._host_threads(workers)
        self.analyzer = DistributionAnalyzer(self.source_dir)
        self.planner: Optional[AugmentationPlanner] = None
        self.manifest_generator: Optional[ManifestGenerator] = None
        self.plan: Dict = {}
        self.tasks: List[dict] = []
        self.completed = self.failed = 0
        self.ranks = _ranks.current()

    @staticmethod
    def _host_threads(requested) -> int:
        """`--workers` keeps the reference's meaning and bounds (default = half the optimal
        count, capped at it; dataset_balancer.py:41-57); here they are decode/encode THREADS."""
        ceiling = get_optimal_worker_count()
        n = max(1, ceiling // 2) if requested is None else max(1, int(requested))
        if n > ceiling:
            logger.warning(f"Requested {n} workers, but only {ceiling} CPUs available; using {ceiling}")
            n = ceiling
        logger.info(f"Using {n} host threads for JPEG decode/encode (max available: {ceiling})")
        return n

    # ------------------------------------------------------------------ planning (host)
    def analyze_distribution(self):
        counts = self.analyzer.analyze()
        self.analyzer.display_distribution()
        return counts

    def calculate_plan(self):
        self.planner = AugmentationPlanner(self.analyzer.counts)
        self.plan = self.planner.calculate_plan()
        return self.plan

    def _fresh_target(self) -> None:
        """The balanced tree starts as a copy of the originals (dataset_balancer.py:70-81)."""
        if not self.source_dir.exists():
            raise FileNotFoundError(f"Source directory not found: {self.source_dir}")
        logger.info(f"Preparing target directory: {self.target_dir}")
        if self.target_dir.exists():
            shutil.rmtree(self.target_dir)
        shutil.copytree(self.source_dir, self.target_dir)

    def _images_by_class(self) -> Dict[str, List[Path]]:
        """class name -> its images in the target tree, `*.JPG` before `*.jpg`, each in glob
        order (the order `random.choice` indexes into; dataset_balancer.py:83-93).  Keyed by
        class name only, like the reference."""
        found: Dict[str, List[Path]] = {}
        for plant in self.target_dir.iterdir():
            if not plant.is_dir():
                continue
            for cls in plant.iterdir():
                if cls.is_dir():
                    found[cls.name] = [*cls.glob("*.JPG"), *cls.glob("*.jpg")]
        return found

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

    # ------------------------------------------------------------------ one share, on one GPU
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

    def _run_group(self, op: str, images: List[np.ndarray], params: List[dict]) -> List[np.ndarray]:
        """One (transform, size) group as one batched launch on this rank's GPU."""
        import torch
        x = torch.from_numpy(np.stack(images)).cuda()
        res = apply_batch(op, x, params)
        if op != "rotate":  # same-sized results: one device->host copy for the group
            host = torch.stack(res).cpu().numpy()
            return [host[j] for j in range(len(images))]
        return [o.cpu().numpy() for o in res]

    def _gpu_stage(self, chunk: List[dict], images: List[Optional[np.ndarray]], pool) -> List[tuple]:
        """Draw every task's parameters (a fresh seeded RNG per task, like the reference's
        `_process_single_transformation`; seeded tasks are independent and drawn on the host
        threads, seed 0 = "unseeded" continues this process's global streams), run the ops
        batched by (transform, size), return (pixels, output_path) pairs for the encoder."""
        groups: Dict[tuple, List[int]] = {}
        params: List[Optional[dict]] = [None] * len(chunk)
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
            groups.setdefault((task["transform_name"], h, w), []).append(k)
        for k, fut in drawing.items():
            params[k] = fut.result()
        out: List[tuple] = []
        for (op, _h, _w), ks in groups.items():
            try:
                res = self._run_group(op, [images[k] for k in ks], [params[k] for k in ks])
                out.extend((r, chunk[k]["output_path"]) for r, k in zip(res, ks))
            except Exception as e:  # noqa: BLE001
                logger.error(f"Failed batch {op}: {e}")
                self.failed += len(ks)
        return out

    def _collect(self, futures) -> None:
        for f in futures:
            if f.result():
                self.completed += 1
            else:
                self.failed += 1

    def _run_share(self, share: List[dict], total: int) -> None:
        """decode | kernels | encode over CHUNK-sized pieces of this rank's share; chunk i+1 is
        being decoded and chunk i-1 encoded while chunk i is on the GPU."""
        chunks = [share[b:b + CHUNK] for b in range(0, len(share), CHUNK)]
        if not chunks:
            return
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            decoding = [pool.submit(self._decode, t) for t in chunks[0]]
            encoding: list = []
            for i, chunk in enumerate(chunks):
                images = [f.result() for f in decoding]
                decoding = [pool.submit(self._decode, t) for t in chunks[i + 1]] if i + 1 < len(chunks) else []
                results = self._gpu_stage(chunk, images, pool)
                self._collect(encoding)
                encoding = [pool.submit(self._encode, r) for r in results]
                done = self.completed + self.failed
                if done and done % 500 < CHUNK:
                    logger.info(f"Progress (rank {self.ranks.rank}): {done}/{len(share)} of this share "
                                f"({total} tasks in all) - {self.completed} success, {self.failed} failed")
            self._collect(encoding)

    # ------------------------------------------------------------------ the whole job
    def execute_balancing(self):
        if not self.plan:
            logger.info("No augmentation plan - skipping execution")
            return
        rk = self.ranks
        if rk.rank == 0:
            self._fresh_target()
            self.tasks = self.build_tasks(self._images_by_class())
        self.tasks = rk.broadcast_object(self.tasks if rk.rank == 0 else None)
        total = len(self.tasks)
        begin, end = contiguous_share(total, rk.rank, rk.world)
        logger.info(f"Starting GPU augmentation: {total} images to generate"
                    + (f" (rank {rk.rank}/{rk.world}: tasks {begin}..{end - 1})" if rk.active else ""))
        streams = (random.getstate(), np.random.get_state())
        self._run_share(self.tasks[begin:end], total)
        random.setstate(streams[0])
        np.random.set_state(streams[1])
        self.completed, self.failed = rk.sum_ints([self.completed, self.failed])
        rk.barrier()  # every share's files are on disk before the tree is listed
        logger.info(f"Augmentation complete: {self.completed} images generated, {self.failed} failed")
        if rk.rank == 0:
            self._generate_augmented_manifest()
        rk.barrier()

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
