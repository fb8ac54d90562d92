"""Thin equivalent of the count/CSV helpers `Augmentation.py` imports from the reference's
Distribution CLI (srcs/cli/Distribution.py:41-86); plots are presentation and out of scope."""
from __future__ import annotations

import csv
from pathlib import Path
from typing import List, Optional, Tuple

IMG_EXTS = {".jpg"}


def count_images(root: Path, plant: Optional[str] = None) -> List[Tuple[str, str, int]]:
    rows: List[Tuple[str, str, int]] = []
    root = Path(root)
    for plant_dir in sorted(d for d in root.iterdir() if d.is_dir()):
        if plant and plant_dir.name != plant:
            continue
        for class_dir in sorted(c for c in plant_dir.iterdir() if c.is_dir()):
            n = sum(1 for f in class_dir.iterdir() if f.is_file() and f.suffix.lower() in IMG_EXTS)
            rows.append((plant_dir.name, class_dir.name, n))
    return rows


def merge_csv(rows: List[Tuple[str, str, int]], csv_path: Path) -> None:
    csv_path = Path(csv_path)
    csv_path.parent.mkdir(parents=True, exist_ok=True)
    with csv_path.open("w", newline="", encoding="utf-8") as f:
        w = csv.writer(f)
        w.writerow(["plant", "class", "count"])
        for r in rows:
            w.writerow(list(r))
