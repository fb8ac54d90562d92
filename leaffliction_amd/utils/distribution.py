"""Counting / CSV helpers of the reference's Distribution CLI (srcs/cli/Distribution.py:22-86),
shared by `Augmentation` and `Distribution`: (plant, class, count) rows for every class folder
that holds at least one `.jpg`, and a distribution CSV that is merged with what is already on
disk (rows for other plants survive a partial re-run)."""
from __future__ import annotations

import csv
import logging
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

IMG_EXTS = {".jpg"}
Row = Tuple[str, str, int]


def count_images(root: Path, plants: Optional[Iterable[str]] = None) -> List[Row]:
    """Sorted (plant, class, n_images) under root/PLANT/CLASS; classes without images are left
    out; `plants` (a name or an iterable of names) restricts the plant folders."""
    if isinstance(plants, str):
        plants = [plants]
    keep = set(plants) if plants else None
    rows: List[Row] = []
    for plant_dir in sorted(d for d in Path(root).iterdir() if d.is_dir()):
        if keep is not None and plant_dir.name not in keep:
            continue
        for class_dir in sorted(c for c in plant_dir.iterdir() if c.is_dir()):
            n = sum(1 for f in class_dir.iterdir() if f.is_file() and f.suffix.lower() in IMG_EXTS)
            if n:
                rows.append((plant_dir.name, class_dir.name, n))
    return rows


def merge_csv(rows: List[Row], csv_path: Path) -> None:
    """Write plant,class,count sorted by (plant, class); counts already in the file are kept
    unless `rows` replaces them; an unreadable or foreign file is recreated."""
    csv_path = Path(csv_path)
    merged: Dict[Tuple[str, str], int] = {}
    if csv_path.exists():
        try:
            with csv_path.open("r", encoding="utf-8") as f:
                reader = csv.DictReader(f)
                if reader.fieldnames and [h.lower() for h in reader.fieldnames] == ["plant", "class", "count"]:
                    for rec in reader:
                        try:
                            merged[(rec["plant"], rec["class"])] = int(rec["count"])
                        except (KeyError, TypeError, ValueError):
                            continue
                else:
                    logging.warning("Replacing incompatible CSV header: %s", csv_path)
        except OSError as exc:
            logging.warning("Unable to read existing CSV (%s), recreating", exc)
    for plant, cls, n in rows:
        merged[(plant, cls)] = n
    csv_path.parent.mkdir(parents=True, exist_ok=True)
    with csv_path.open("w", newline="", encoding="utf-8") as f:
        w = csv.writer(f)
        w.writerow(["plant", "class", "count"])
        for plant, cls in sorted(merged):
            w.writerow([plant, cls, merged[(plant, cls)]])


def plot_per_plant(rows: List[Row], out_dir: Path) -> bool:
    """One bar and one pie chart per plant (PLANT_bar.png / PLANT_pie.png).  Presentation only:
    skipped (returns False) when matplotlib is not installed."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception as exc:  # noqa: BLE001
        logging.warning("matplotlib unavailable, skipping plots (%s)", exc)
        return False
    by_plant: Dict[str, List[Tuple[str, int]]] = {}
    for plant, cls, n in rows:
        by_plant.setdefault(plant, []).append((cls, n))
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    for plant, pairs in by_plant.items():
        names, values = [c for c, _ in pairs], [n for _, n in pairs]
        fig = plt.figure()
        plt.title(f"Distribution — {plant} (bar)")
        plt.bar(names, values)
        plt.xlabel("Class")
        plt.ylabel("Images")
        plt.xticks(rotation=45, ha="right")
        fig.tight_layout()
        fig.savefig(str(out_dir / f"{plant}_bar.png"), dpi=150)
        plt.close(fig)
        fig = plt.figure()
        plt.title(f"Distribution — {plant} (pie)")
        plt.pie(values, labels=names, autopct="%1.1f%%")
        fig.tight_layout()
        fig.savefig(str(out_dir / f"{plant}_pie.png"), dpi=150)
        plt.close(fig)
    return True
