"""Worker-count heuristic (mirror of srcs/utils/system_info.py:9-46)."""
from __future__ import annotations

import os
import platform


def get_available_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def get_optimal_worker_count() -> int:
    cores = get_available_cores()
    if cores <= 2:
        return 1
    if cores <= 4:
        return max(1, cores - 1)
    if platform.system() == "Darwin" and platform.machine() == "arm64":
        return min(8, cores)
    return max(1, int(cores * 0.75))
