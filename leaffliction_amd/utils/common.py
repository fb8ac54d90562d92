"""Logging helpers (mirror of srcs/utils/common.py:15-48, plain formatter)."""
from __future__ import annotations

import logging


def setup_logging(level: int = logging.INFO) -> None:
    root = logging.getLogger()
    if not root.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s | %(levelname)s | %(message)s",
                                         "%Y-%m-%d %H:%M:%S"))
        root.addHandler(h)
    root.setLevel(level)


def get_logger(name: str) -> logging.Logger:
    return logging.getLogger(name)
