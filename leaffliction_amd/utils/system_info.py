"""Worker-count heuristic (mirror of srcs/utils/system_info.py:9-46)."""
from __future__ import annotations

import os
import platform


def get_available_cores() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a container shows every core of its host but is granted a share of them; sizing a worker
    pool by the host count oversubscribes the share and pays hundreds of process spawns), divided
    by the ranks of this node when running one process per GPU (LOCAL_WORLD_SIZE)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    # one process per GPU under torch.distributed.run: the ranks of a node share its cores
    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        local_world = 1
    if local_world > 1:
        n = max(1, n // local_world)
    return n


def get_optimal_worker_count() -> int:
    cores = get_available_cores()
    if cores <= 2:
        return 1
    if cores <= 4:
        return max(1, cores - 1)
    if platform.system() == "Darwin" and platform.machine() == "arm64":
        return min(8, cores)
    return max(1, int(cores * 0.75))


def cap_torch_threads() -> None:
    """torch sizes its intra-op pool by the cores it can SEE (256 on a box that grants 16): every small CPU tensor
    op of a host loop then pays a 256-thread fork/join (the per-step dropout draws of `fit`: 19-27 ms, as long
    as the whole bf16 step).  Never more threads than the process may use; called when the library is loaded."""
    import torch
    cores = get_available_cores()
    if torch.get_num_threads() > cores:
        torch.set_num_threads(cores)
