"""What the sharded (no-exchange) paths need from `torch.distributed`: augmentation and batch
inference partition independent units — tasks, files — into one contiguous share per GPU
(SURVEY §8e), so the only cross-rank traffic is control: the unit list goes out from rank 0,
integer counters are summed, results come back in input order, and a barrier precedes the files
rank 0 writes.  No data-path collective.

`current()` returns the live process group (initialised by the CLI under
`python -m torch.distributed.run`, backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests)
or a single-process stand-in with the same methods.
"""
from __future__ import annotations

import os
from typing import Any, List, Sequence, Tuple


def contiguous_share(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of rank's share when `total` units are dealt out in contiguous runs whose
    lengths differ by at most one (the first `total % world` ranks get the longer runs)."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class Solo:
    rank, world, active = 0, 1, False

    def barrier(self) -> None: ...

    def broadcast_object(self, obj: Any, src: int = 0) -> Any:
        return obj

    def sum_ints(self, vals: Sequence[int]) -> List[int]:
        return [int(v) for v in vals]

    def gather_in_order(self, part: list) -> list:
        return list(part)


class Ranks:
    def __init__(self, dist) -> None:
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.active = self.world > 1

    def barrier(self) -> None:
        self.dist.barrier()

    def broadcast_object(self, obj: Any, src: int = 0) -> Any:
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def sum_ints(self, vals: Sequence[int]) -> List[int]:
        import torch
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(list(vals), dtype=torch.int64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def gather_in_order(self, part: list) -> list:
        """Concatenation of every rank's list in rank order (contiguous shares => input order),
        on every rank."""
        parts: List[Any] = [None] * self.world
        self.dist.all_gather_object(parts, part)
        return [x for p in parts for x in p]


def current():
    try:
        import torch.distributed as dist
    except ImportError:
        return Solo()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return Ranks(dist)
    return Solo()


def init_from_env():
    """Join the process group `torch.distributed.run` describes (WORLD_SIZE > 1), one GPU per
    rank; a plain `python -m ...` run stays single-process.  LEAFFLICTION_DIST_BACKEND=gloo lets
    more ranks than GPUs share the cards (rehearsals, tests).  Returns current()."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return Solo()
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        backend = os.environ.get("LEAFFLICTION_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        kw = {}
        if torch.cuda.is_available():
            dev = local if backend == "nccl" else local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if os.environ.get("LEAFFLICTION_DIST_TIMEOUT"):   # seconds; the default (30 min) turns a lost rank into a stall
            from datetime import timedelta
            kw["timeout"] = timedelta(seconds=float(os.environ["LEAFFLICTION_DIST_TIMEOUT"]))
        dist.init_process_group(backend, rank=int(os.environ.get("RANK", "0")), world_size=world, **kw)
    return current()
