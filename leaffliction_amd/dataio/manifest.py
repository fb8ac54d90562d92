"""Manifest contract (mirror of srcs/dataio/manifest.py:10-42).

`manifest_*.json` = {"meta": {...}, "items": [{id, plant, class, label, split, src}]};
label indices are the rank of each label among the sorted unique TRAIN labels — that
mapping is part of the artifact contract (labels.json) and must be bit-exact.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Iterable, List


@dataclass(frozen=True)
class ManifestItem:
    id: str
    plant: str
    cls: str
    label: str
    split: str
    src: Path


def load_manifest(path: Path) -> List[ManifestItem]:
    with Path(path).open("r", encoding="utf-8") as f:
        doc = json.load(f)
    out: List[ManifestItem] = []
    for entry in doc["items"]:
        out.append(ManifestItem(id=entry["id"], plant=entry["plant"], cls=entry["class"],
                                label=entry["label"], split=entry["split"],
                                src=Path(entry["src"])))
    return out


def select_items(items: Iterable[ManifestItem], split: str) -> List[ManifestItem]:
    return [it for it in items if it.split == split]


def build_label_mapping(train_items: List[ManifestItem]) -> Dict[str, int]:
    return {label: idx for idx, label in enumerate(sorted({it.label for it in train_items}))}
