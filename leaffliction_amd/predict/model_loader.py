"""meta.json + leaf_cnn.keras loader (mirror of srcs/predict/model_loader.py:12-59)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List

from ..utils.common import get_logger

logger = get_logger(__name__)


class ModelLoader:
    def __init__(self, learnings_dir):
        self.learnings_dir = Path(learnings_dir)
        self.meta_data: Dict[str, Any] = {}
        self.model = None

    def load(self):
        meta_path = self.learnings_dir / "meta.json"
        if not meta_path.exists():
            raise FileNotFoundError(f"Meta file not found: {meta_path}")
        with open(meta_path, "r", encoding="utf-8") as f:
            self.meta_data = json.load(f)
        model_file = self.meta_data.get("model_file")
        if not model_file:
            raise ValueError("Model file not specified in metadata")
        model_path = Path(model_file)  # cwd-relative, like the reference (SURVEY B-13)
        if not model_path.exists():
            raise FileNotFoundError(f"Model file not found: {model_path}")
        from ..model.cnn import load_model
        self.model = load_model(model_path)
        logger.info("Model and metadata loaded successfully")

    @property
    def labels(self) -> List[str]:
        return self.meta_data.get("labels", [])

    @property
    def img_size(self) -> int:
        return self.meta_data.get("data", {}).get("img_size", 224)

    @property
    def num_classes(self) -> int:
        return len(self.labels)
