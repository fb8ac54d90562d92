"""The manifest artifact (`manifest_*.json`) as the loaders see it.

Schema (srcs/dataio/manifest.py:10-42): `{"meta": {...}, "items": [...]}` with
one record per image: `id`, `plant`, `class`, `label` (= "<plant>__<class>"), `split`
("train" / "val") and `src` (path of the JPEG).  Label indices are ranks among the sorted unique
labels of the TRAIN items; that mapping is written to labels.json and read back by predict, so it
has to come out identical to the reference's.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Iterable, List, NamedTuple


class ManifestItem(NamedTuple):
    id: str
    plant: str
    cls: str      # the record's "class" key (a Python keyword)
    label: str
    split: str
    src: Path

    @classmethod
    def from_record(cls, rec: dict) -> "ManifestItem":
        return cls(rec["id"], rec["plant"], rec["class"], rec["label"], rec["split"], Path(rec["src"]))


def load_manifest(path: Path) -> List[ManifestItem]:
    records = json.loads(Path(path).read_text(encoding="utf-8"))["items"]
    return [ManifestItem.from_record(r) for r in records]


def select_items(items: Iterable[ManifestItem], split: str) -> List[ManifestItem]:
    return [it for it in items if it.split == split]


def build_label_mapping(train_items: List[ManifestItem]) -> Dict[str, int]:
    ordered = sorted(set(it.label for it in train_items))
    return dict(zip(ordered, range(len(ordered))))
