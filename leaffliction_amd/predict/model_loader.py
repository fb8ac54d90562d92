"""Reads what `train` left in the learnings directory: meta.json and the model file it names
(the contract of srcs/predict/model_loader.py:12-59).  `meta["model_file"]` is resolved against
the current working directory, as the reference does."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List

from ..utils.common import get_logger

logger = get_logger(__name__)


def _read_meta(learnings_dir: Path) -> Dict[str, Any]:
    meta_path = learnings_dir / "meta.json"
    if not meta_path.exists():
        raise FileNotFoundError(f"Meta file not found: {meta_path}")
    return json.loads(meta_path.read_text(encoding="utf-8"))


def _model_path(meta: Dict[str, Any]) -> Path:
    name = meta.get("model_file")
    if not name:
        raise ValueError("Model file not specified in metadata")
    path = Path(name)
    if not path.exists():
        raise FileNotFoundError(f"Model file not found: {path}")
    return path


class ModelLoader:
    def __init__(self, learnings_dir):
        self.learnings_dir = Path(learnings_dir)
        self.meta_data: Dict[str, Any] = {}
        self.model = None

    def load(self):
        from ..model.cnn import load_model
        self.meta_data = _read_meta(self.learnings_dir)
        self.model = load_model(_model_path(self.meta_data))
        logger.info("Model and metadata loaded successfully")

    labels = property(lambda self: list(self.meta_data.get("labels", [])))
    img_size = property(lambda self: int(self.meta_data.get("data", {}).get("img_size", 224)))
    num_classes = property(lambda self: len(self.labels))
