"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in CPU tests).  The training path has exactly one exchange step per
optimisation step: an all-reduce (sum) of the flat gradient bucket; BatchNorm statistics
stay per-GPU (the reference has no multi-device semantics to match).
Validation counts (confusion matrices, correct/total) are integer all-reduces (exact)."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


class DataParallel:
    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None) -> None:
        import torch.distributed as dist
        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.active = self.world > 1
        # "f32" (default): the flat gradient bucket crosses xGMI as it is (5.0 MB, exact sum);
        # "bf16": rounded to bf16 for the all-reduce (2.5 MB) and widened again — BASELINE.md section 4
        self.bucket_dtype = os.environ.get("LEAFFLICTION_GRAD_BUCKET", "f32")
        self._bucket16: Optional[torch.Tensor] = None
        # overlap the exchange with the backward pass (LeafCNN.train_step, `grad_overlap`): the bucket goes out in two
        # pieces, the first while the 224 x 224 layers' backward still computes.  LEAFFLICTION_GRAD_OVERLAP=0: one
        # all-reduce after the whole backward pass (round 1 / 2 behaviour; same bits).
        self.overlap = self.world > 1 and os.environ.get("LEAFFLICTION_GRAD_OVERLAP", "1") != "0"
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            if os.environ.get("LEAFFLICTION_DIST_TIMEOUT"):
                from datetime import timedelta
                kw["timeout"] = timedelta(seconds=float(os.environ["LEAFFLICTION_DIST_TIMEOUT"]))
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)

    # -- the one exchange step of the training path
    def allreduce_grads(self, flat_g: torch.Tensor) -> torch.Tensor:
        """Sum the flat gradient bucket over ranks (local gradients are already scaled by
        1/global_batch, so the sum is the global-batch mean gradient)."""
        if not self.active:
            return flat_g
        if self.bucket_dtype == "bf16" and flat_g.is_cuda:
            from .. import nn
            if self._bucket16 is None or self._bucket16.numel() != flat_g.numel():
                self._bucket16 = torch.empty(flat_g.numel(), dtype=torch.bfloat16, device=flat_g.device)
            nn.cast_f32_bf16(flat_g, self._bucket16)
            self.dist.all_reduce(self._bucket16, op=self.dist.ReduceOp.SUM)
            nn.cast_bf16_f32(self._bucket16, flat_g)
        else:
            self.dist.all_reduce(flat_g, op=self.dist.ReduceOp.SUM)
        return flat_g

    def allreduce_begin(self, flat_g: torch.Tensor, lo: int, hi: int):
        """Start the sum of flat_g[lo:hi] over the ranks without waiting for it: the collective runs on the
        backend's own stream behind everything issued to the current stream so far, and beside what is issued
        next (RCCL over xGMI while the rest of the backward pass computes).  Returns a handle for allreduce_finish."""
        if not self.active or hi <= lo:
            return None
        part = flat_g[lo:hi]
        if self.bucket_dtype == "bf16" and flat_g.is_cuda:
            from .. import nn
            if self._bucket16 is None or self._bucket16.numel() != flat_g.numel():
                self._bucket16 = torch.empty(flat_g.numel(), dtype=torch.bfloat16, device=flat_g.device)
            b16 = self._bucket16[lo:hi]
            nn.cast_f32_bf16(part, b16)
            return (self.dist.all_reduce(b16, op=self.dist.ReduceOp.SUM, async_op=True), b16, part)
        return (self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM, async_op=True), None, part)

    def allreduce_finish(self, flat_g: torch.Tensor, handles) -> torch.Tensor:
        """The current stream waits for the exchanges started by allreduce_begin (no host wait with RCCL); the bf16
        bucket's pieces are widened back into the fp32 gradient."""
        for hd in handles:
            if hd is None:
                continue
            work, b16, part = hd
            work.wait()
            if b16 is not None:
                from .. import nn
                nn.cast_bf16_f32(b16, part)
        return flat_g

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.active:
            self.dist.broadcast(t, src)
        return t

    def allreduce_counts(self, counts: torch.Tensor) -> torch.Tensor:
        """Integer counters (confusion matrix, correct/total): exact sum over ranks."""
        if self.active:
            self.dist.all_reduce(counts, op=self.dist.ReduceOp.SUM)
        return counts

    def allreduce_scalars(self, vals: List[float]) -> List[float]:
        if not self.active:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def barrier(self) -> None:
        if self.active:
            self.dist.barrier()

    def shutdown(self) -> None:
        if self.active and self.dist.is_initialized():
            self.dist.destroy_process_group()


def shard_slice(indexes: List[int], rank: int, world: int) -> List[int]:
    """Rank-strided slice of one global batch (same permutation on every rank)."""
    return indexes[rank::world]


def split_counts(total: int, world: int) -> Tuple[int, ...]:
    return tuple(len(range(r, total, world)) for r in range(world))
