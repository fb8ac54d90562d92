"""Distribution analysis, augmentation plan and augmented-manifest writer.

Mirror of srcs/preprocessing/dataset_components.py:12-187.  Semantics kept, including
the quirks SURVEY Appendix B lists: directory order is `iterdir()` order (unsorted),
deficits are keyed by class name only, every manifest item is split="train" and
"augmentation_seed" is the literal 42.
"""
from __future__ import annotations

import json
from datetime import datetime, timezone
from pathlib import Path
from typing import Dict

from ..utils.common import get_logger

logger = get_logger(__name__)

TRANSFORMATIONS = ["flip", "rotate", "skew", "shear", "crop", "distortion"]


class DistributionAnalyzer:
    IMG_EXTS = {".jpg"}

    def __init__(self, input_path):
        self.input_path = Path(input_path)
        self.counts: Dict[str, Dict[str, int]] = {}
        self.original_manifest = None

    def _count_dir(self, root: Path) -> Dict[str, Dict[str, int]]:
        if not root.exists():
            raise FileNotFoundError(f"Dataset directory not found: {root}")
        counts: Dict[str, Dict[str, int]] = {}
        for plant_dir in root.iterdir():
            if not plant_dir.is_dir():
                continue
            for class_dir in plant_dir.iterdir():
                if not class_dir.is_dir():
                    continue
                n = sum(1 for f in class_dir.iterdir()
                        if f.is_file() and f.suffix.lower() in self.IMG_EXTS)
                if n > 0:
                    per_plant = counts.setdefault(plant_dir.name, {})
                    per_plant[class_dir.name] = per_plant.get(class_dir.name, 0) + n
        return counts

    def _count_manifest(self, path: Path) -> Dict[str, Dict[str, int]]:
        with path.open("r", encoding="utf-8") as f:
            manifest = json.load(f)
        self.original_manifest = manifest
        counts: Dict[str, Dict[str, int]] = {}
        for item in manifest.get("items", []):
            plant, cls = item.get("plant"), item.get("class")
            if plant and cls:
                per_plant = counts.setdefault(plant, {})
                per_plant[cls] = per_plant.get(cls, 0) + 1
        return counts

    def analyze(self):
        if not self.input_path.exists():
            raise FileNotFoundError(f"Input not found: {self.input_path}")
        if self.input_path.is_dir():
            self.counts = self._count_dir(self.input_path)
        else:
            self.counts = self._count_manifest(self.input_path)
        return self.counts

    def display_distribution(self):
        logger.info("Analyzing dataset distribution...")
        for plant, classes in sorted(self.counts.items()):
            logger.info(f"\n[{plant}]")
            for class_name, count in sorted(classes.items()):
                logger.info(f"  {class_name}: {count} images")


class AugmentationPlanner:
    def __init__(self, counts):
        self.counts = counts
        self.plan = {}

    def calculate_plan(self):
        """deficit = plant max - count; split over the six transforms, remainder to the first."""
        deficits: Dict[str, int] = {}
        for _plant, classes in self.counts.items():
            plant_max = max(classes.values())
            for class_name, count in classes.items():
                if plant_max - count > 0:
                    deficits[class_name] = plant_max - count
        if not deficits:
            logger.info("Dataset already balanced - no augmentations needed")
            return {}
        plan: Dict[str, Dict[str, int]] = {}
        for class_name, deficit in deficits.items():
            base, rem = divmod(deficit, len(TRANSFORMATIONS))
            per = {}
            for i, name in enumerate(TRANSFORMATIONS):
                cnt = base + (1 if i < rem else 0)
                if cnt > 0:
                    per[name] = cnt
            plan[class_name] = per
        self.plan = plan
        for class_name, deficit in sorted(deficits.items()):
            logger.info(f"  Class: {class_name} - {deficit} images needed")
        return plan


class ManifestGenerator:
    def __init__(self, original_manifest, source_dir, target_dir, workers):
        self.original_manifest = original_manifest
        self.source_dir = Path(source_dir)
        self.target_dir = Path(target_dir)
        self.workers = workers

    def generate_augmented_manifest(self):
        items = []
        for plant_dir in self.target_dir.iterdir():
            if not plant_dir.is_dir():
                continue
            for class_dir in plant_dir.iterdir():
                if not class_dir.is_dir():
                    continue
                for img in class_dir.iterdir():
                    if not img.is_file():
                        continue
                    items.append({
                        "plant": plant_dir.name,
                        "class": class_dir.name,
                        "label": f"{plant_dir.name}__{class_dir.name}",
                        "split": "train",
                        "src": str(img),
                        "id": str(img.relative_to(self.target_dir)),
                        "augmented": "_aug_" in img.stem,
                    })
        created_at = original_seed = None
        if isinstance(self.original_manifest, dict):
            meta = self.original_manifest.get("meta", {})
            created_at, original_seed = meta.get("created_at"), meta.get("seed")
        n_aug = sum(1 for i in items if i["augmented"])
        return {
            "meta": {
                "created_at": created_at,
                "augmented_at": datetime.now(timezone.utc).isoformat(),
                "original_seed": original_seed,
                "augmentation_seed": 42,
                "workers": self.workers,
                "src_root": str(self.target_dir),
                "total_images": len(items),
                "original_images": len(items) - n_aug,
                "augmented_images": n_aug,
            },
            "items": items,
        }

    def save_manifest(self, manifest, output_path):
        with open(output_path, "w", encoding="utf-8") as f:
            json.dump(manifest, f, indent=2, ensure_ascii=False)
        logger.info(f"Augmented manifest saved: {output_path}")
