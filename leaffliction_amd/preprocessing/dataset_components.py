"""Counting, planning and manifest writing around the augmentation loop.

Contract of srcs/preprocessing/dataset_components.py:12-187, quirks included:
class folders are visited in `iterdir()` order (unsorted, so the plan's dict order follows the file
system); deficits and plans are keyed by class name alone; every item of the augmented manifest is
`split: "train"`, and its meta carries the literal `"augmentation_seed": 42`.
"""
from __future__ import annotations

import json
import os
from datetime import datetime, timezone
from pathlib import Path
from typing import Dict, Iterator, Tuple

from ..utils.common import get_logger

logger = get_logger(__name__)

TRANSFORMATIONS = ["flip", "rotate", "skew", "shear", "crop", "distortion"]
Counts = Dict[str, Dict[str, int]]


def _class_dirs(root: Path) -> Iterator[Tuple[Path, Path]]:
    """(plant_dir, class_dir) pairs under root/PLANT/CLASS in directory order."""
    for plant_dir in root.iterdir():
        if plant_dir.is_dir():
            for class_dir in plant_dir.iterdir():
                if class_dir.is_dir():
                    yield plant_dir, class_dir


def _bump(counts: Counts, plant: str, cls: str, n: int) -> None:
    row = counts.setdefault(plant, {})
    row[cls] = row.get(cls, 0) + n


class DistributionAnalyzer:
    IMG_EXTS = {".jpg"}

    def __init__(self, input_path):
        self.input_path = Path(input_path)
        self.counts: Counts = {}
        self.original_manifest = None

    def _is_image(self, f: Path) -> bool:
        return f.is_file() and f.suffix.lower() in self.IMG_EXTS

    def _count_dir(self, root: Path) -> Counts:
        if not root.exists():
            raise FileNotFoundError(f"Dataset directory not found: {root}")
        counts: Counts = {}
        for plant_dir, class_dir in _class_dirs(root):
            n = sum(map(self._is_image, class_dir.iterdir()))
            if n:
                _bump(counts, plant_dir.name, class_dir.name, n)
        return counts

    def _count_manifest(self, path: Path) -> Counts:
        self.original_manifest = json.loads(path.read_text(encoding="utf-8"))
        counts: Counts = {}
        for rec in self.original_manifest.get("items", []):
            if rec.get("plant") and rec.get("class"):
                _bump(counts, rec["plant"], rec["class"], 1)
        return counts

    def analyze(self):
        src = self.input_path
        if not src.exists():
            raise FileNotFoundError(f"Input not found: {src}")
        self.counts = self._count_dir(src) if src.is_dir() else self._count_manifest(src)
        return self.counts

    def display_distribution(self):
        logger.info("Analyzing dataset distribution...")
        for plant in sorted(self.counts):
            logger.info(f"\n[{plant}]")
            for class_name in sorted(self.counts[plant]):
                logger.info(f"  {class_name}: {self.counts[plant][class_name]} images")


class AugmentationPlanner:
    def __init__(self, counts):
        self.counts = counts
        self.plan = {}

    def calculate_plan(self):
        """Per class: deficit to the largest class of its plant, dealt over the six transforms in
        TRANSFORMATIONS order (`deficit // 6` each, the first `deficit % 6` get one more; zero
        shares are left out).  Classes that need nothing do not appear."""
        need: Dict[str, int] = {}
        for classes in self.counts.values():
            top = max(classes.values())
            need.update({name: top - n for name, n in classes.items() if n < top})
        if not need:
            logger.info("Dataset already balanced - no augmentations needed")
            return {}
        k = len(TRANSFORMATIONS)
        self.plan = {
            name: {t: deficit // k + (i < deficit % k) for i, t in enumerate(TRANSFORMATIONS)
                   if deficit // k + (i < deficit % k) > 0}
            for name, deficit in need.items()
        }
        for name in sorted(need):
            logger.info(f"  Class: {name} - {need[name]} images needed")
        return self.plan


class ManifestGenerator:
    def __init__(self, original_manifest, source_dir, target_dir, workers):
        self.original_manifest = original_manifest
        self.source_dir = Path(source_dir)
        self.target_dir = Path(target_dir)
        self.workers = workers

    def _records(self) -> Iterator[dict]:
        for plant_dir, class_dir in _class_dirs(self.target_dir):
            plant, cls = plant_dir.name, class_dir.name
            base, rel, label = str(class_dir) + os.sep, plant + os.sep + cls + os.sep, f"{plant}__{cls}"
            with os.scandir(class_dir) as entries:   # directory order, as Path.iterdir(); no stat per file
                for e in entries:
                    if e.is_file():
                        name = e.name
                        yield {"plant": plant, "class": cls, "label": label, "split": "train",
                               "src": base + name, "id": rel + name,
                               "augmented": "_aug_" in os.path.splitext(name)[0]}

    def generate_augmented_manifest(self):
        items = list(self._records())
        n_aug = sum(rec["augmented"] for rec in items)
        upstream = self.original_manifest.get("meta", {}) if isinstance(self.original_manifest, dict) else {}
        meta = {"created_at": upstream.get("created_at"),
                "augmented_at": datetime.now(timezone.utc).isoformat(),
                "original_seed": upstream.get("seed"),
                "augmentation_seed": 42,
                "workers": self.workers,
                "src_root": str(self.target_dir),
                "total_images": len(items),
                "original_images": len(items) - n_aug,
                "augmented_images": n_aug}
        return {"meta": meta, "items": items}

    @staticmethod
    def _dumps(manifest) -> str:
        """`json.dumps(manifest, indent=2, ensure_ascii=False)`, byte for byte — which is what the reference
        writes (dataset_components.py save_manifest) — without the pure-Python encoder an `indent` selects: the
        flat item records go through the C encoder one call each, with the indentation in the separator."""
        items = manifest.get("items") if isinstance(manifest, dict) else None
        flat = (str, int, float, bool, type(None))
        if (not isinstance(items, list) or not items or list(manifest)[-1] != "items" or len(manifest) < 2
                or not all(isinstance(r, dict) and r and all(isinstance(k, str) and isinstance(v, flat)
                                                             for k, v in r.items()) for r in items)):
            return json.dumps(manifest, indent=2, ensure_ascii=False)
        head = json.dumps({k: v for k, v in manifest.items() if k != "items"}, indent=2, ensure_ascii=False)
        sep = (",\n      ", ": ")
        recs = ["    {\n      " + json.dumps(r, ensure_ascii=False, separators=sep)[1:-1] + "\n    }" for r in items]
        return head[:-2] + ',\n  "items": [\n' + ",\n".join(recs) + "\n  ]\n}"

    def write_augmented_manifest(self, output_path) -> int:
        """`save_manifest(generate_augmented_manifest(), output_path)` without the record dictionaries in between: the
        same bytes (tests/test_host_logic.py), the text of each record put together from its class's constant lines and
        the two strings that vary (the C encoder escapes them).  Returns the number of items."""
        parts, n_items, n_aug = [], 0, 0
        enc = json.JSONEncoder(ensure_ascii=False).encode
        for plant_dir, class_dir in _class_dirs(self.target_dir):
            plant, cls = plant_dir.name, class_dir.name
            base, rel = str(class_dir) + os.sep, plant + os.sep + cls + os.sep
            head = ('    {\n      "plant": ' + enc(plant) + ',\n      "class": ' + enc(cls) + ',\n      "label": ' +
                    enc(f"{plant}__{cls}") + ',\n      "split": "train",\n      "src": ')
            with os.scandir(class_dir) as entries:
                for e in entries:
                    if e.is_file():
                        name = e.name
                        aug = "_aug_" in os.path.splitext(name)[0]
                        n_aug += aug
                        parts.append(head + enc(base + name) + ',\n      "id": ' + enc(rel + name) +
                                     (',\n      "augmented": true\n    }' if aug else ',\n      "augmented": false\n    }'))
            n_items = len(parts)
        if not parts:   # an empty list is written by the general encoder
            self.save_manifest(self.generate_augmented_manifest(), output_path)
            return 0
        upstream = self.original_manifest.get("meta", {}) if isinstance(self.original_manifest, dict) else {}
        meta = {"created_at": upstream.get("created_at"),
                "augmented_at": datetime.now(timezone.utc).isoformat(),
                "original_seed": upstream.get("seed"),
                "augmentation_seed": 42,
                "workers": self.workers,
                "src_root": str(self.target_dir),
                "total_images": n_items,
                "original_images": n_items - n_aug,
                "augmented_images": n_aug}
        head = json.dumps({"meta": meta}, indent=2, ensure_ascii=False)
        Path(output_path).write_text(head[:-2] + ',\n  "items": [\n' + ",\n".join(parts) + "\n  ]\n}", encoding="utf-8")
        logger.info(f"Augmented manifest saved: {output_path}")
        return n_items

    def save_manifest(self, manifest, output_path):
        Path(output_path).write_text(self._dumps(manifest), encoding="utf-8")
        logger.info(f"Augmented manifest saved: {output_path}")
