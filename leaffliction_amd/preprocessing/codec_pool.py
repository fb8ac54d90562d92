"""The JPEG edge of the balancer in worker PROCESSES, around the GPU stage.

The reference decodes, transforms and encodes each task inside a pool process
(srcs/preprocessing/dataset_balancer.py:137-141,201-207).  Here the transform runs on the GPU, but
the two codec halves are still libjpeg on host cores, and they are the whole cost of the job
(≈0.5 ms decode + ≈0.7 ms encode per 224x224 image against microseconds of kernels).  Python
threads cannot feed that: 16 threads reached 1.8 k images/s on a box whose 16 cores run the
reference's own process pool at 10.8 k images/s.  So the codec work goes to processes as well, and
pixels cross the process boundary through shared memory, never through pickles:

  decode worker: JPEG -> RGB uint8 written straight into the task's slot of the INPUT slab — or, for a
      baseline 4:2:0 file of whole MCUs when the GPU path is on, only its markers (tables + the un-stuffed scan into
      the slot: Huffman decoding, IDCT, upsampling and colour conversion are the GPU's) or its markers and Huffman
      decoding (tables + coefficients into the slot); the
      task's random parameters are drawn there too (a fresh seeded RNG per task, exactly what the
      reference's worker does); the distortion's 150 k normal deviates are either left to the GPU (the worker hands
      the seed on) or made here, cast to uint8 by numpy's own rule (image_augmenter.py:121-123), into the NOISE slab;
      the workers also copy the originals into the target tree, a batch of files per job;
  main process:  one H2D per (transform, size) group, batched kernels, D2H into the OUTPUT slab;
  encode worker: slot of the OUTPUT slab -> JPEG quality 95 -> the task's output path.  Images of whole
      16x16 MCUs arrive as finished JPEG scans (colour conversion, 4:2:0 downsampling, DCT, quantisation,
      Huffman coding and byte stuffing ran on the GPU: ops.jpeg_fdct_quant_u8, ops.jpeg_entropy_u8) and the
      worker only adds the markers: the same bytes Pillow would write.

Only paths, shapes, seeds and a few floats are pickled.  Slots are fixed-size (an image that does
not fit — a rotated copy of an unusually large original — falls back to a pickled array).
"""
from __future__ import annotations

import random
from concurrent.futures import Future, ProcessPoolExecutor
from multiprocessing import get_context, shared_memory
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

_SLABS: Dict[str, shared_memory.SharedMemory] = {}
_DEBUG: list = []


def _slabs(names: Dict[str, str]) -> Dict[str, shared_memory.SharedMemory]:
    """Attach (once per worker) to the parent's slabs."""
    for key, name in names.items():
        cur = _SLABS.get(key)
        if cur is None or cur.name != name:
            _SLABS[key] = shared_memory.SharedMemory(name=name)
    return _SLABS


def _warm(_i: int) -> bool:
    """First job of every worker: pay the imports while the parent is still copying the originals."""
    from ..utils.image_utils import ImageLoader  # noqa: F401
    from .image_augmenter import draw_params  # noqa: F401
    from ..utils import jpeg_host
    jpeg_host.load()
    import os
    import time
    stacks = os.environ.get("LEAFFLICTION_DEBUG_STACKS")
    if stacks and not _DEBUG:   # diagnosis of a stuck job: every worker leaves its Python stacks after 100 s
        import faulthandler
        _DEBUG.append(open(os.path.join(stacks, f"codec_{os.getppid()}_{os.getpid()}.txt"), "w"))
        faulthandler.dump_traceback_later(100, exit=False, file=_DEBUG[0])
    time.sleep(0.05)   # long enough that every worker of the pool takes one
    return True


def touch_files(paths: Sequence[str]) -> int:
    """Empty files, in order (the balancer lays the target tree out before the bytes follow)."""
    import os
    for dst in paths:
        os.close(os.open(dst, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666))
    return len(paths)


def copy_files(pairs: Sequence[tuple]) -> int:
    """shutil.copy2 for a batch of (source, destination) pairs: the balancer's copy of the originals
    (dataset_balancer.py:70-81), done by the codec workers between their other jobs."""
    import shutil
    for src, dst in pairs:
        shutil.copy2(src, dst)
    return len(pairs)


def _decode_jobs(names: Dict[str, str], jobs: Sequence[tuple]):
    """jobs: (source path, transform, seed, input offset, noise offset, slot bytes[, want coefficients: 0 | 1 | 2
    [, noise on the GPU]]).
    Returns per job ("ok", shape, params) | ("coef", shape, params) | ("scan", shape, params) | ("big", array, params)
    | ("err", message).
    "coef": the file was a baseline 4:2:0 JPEG of whole MCUs and the slot holds its quantisation tables and
    Huffman-decoded coefficients (libleafcodec.so); the GPU finishes the decoding (ops.jpeg_idct_rgb_u8).
    "scan" (asked for with 2): only the file's markers were read here; the slot holds the tables and the un-stuffed
    scan, and the Huffman decoding is the GPU's as well (ops.jpeg_huffman_u8); shape = (h, w, 3, bytes of the slot in use)."""
    from ..utils import jpeg_host
    from ..utils.image_utils import ImageLoader, _checked
    from .image_augmenter import draw_params
    out = []
    slabs = _slabs(names)
    buf_in, buf_noise = slabs["in"].buf, slabs["noise"].buf
    for job in jobs:
        path, op, seed, off, noff, cap = job[:6]
        try:
            arr, coef_hw, scan_hw = None, None, None
            if len(job) > 6 and job[6]:
                with open(_checked(path, what="Image"), "rb") as f:   # the loader's own checks (image_utils.py:19-33)
                    data = f.read()
                dst = np.frombuffer(buf_in, np.uint8, cap, off)
                if job[6] == 2:
                    got = jpeg_host.scan_prepare_into(data, dst)
                    scan_hw = (got[0], got[1], got[3]) if got is not None else None
                if scan_hw is None:
                    coef_hw = jpeg_host.read_file_into(data, dst)
            if scan_hw is not None:
                h, w = scan_hw[:2]
            elif coef_hw is None:
                arr = ImageLoader.load_as_array(path)
                h, w, _c = arr.shape
            else:
                h, w = coef_hw
            params = None
            if seed and op == "distortion" and 0 < seed < 2 ** 32:
                # RandomState(seed).normal(0, 5, (h, w, 3)).astype(uint8) and then random.Random(seed).uniform(0, 2)
                # (image_augmenter.py:121-127): the noise plane comes from libleafcodec's restatement of numpy's
                # legacy stream (bit-equal, tests/test_jpeg_codec.py; 2.6 ms against numpy's 3.4 ms for 224 x 224 x 3)
                from .image_augmenter import NOISE_LEVEL
                params = {"cutoff": random.Random(seed).uniform(0, 2)}
                nbytes = h * w * 3
                if len(job) > 7 and job[7]:
                    params["noise_seed"] = seed   # the parent has the plane made on the GPU (ops.legacy_normal_u8)
                elif nbytes <= cap:
                    jpeg_host.legacy_normal_u8(seed, 0.0, float(NOISE_LEVEL), np.frombuffer(buf_noise, np.uint8, nbytes, noff))
                    params["noise8"] = None      # in the noise slab, at this task's slot
                else:
                    n8 = np.empty((h, w, 3), np.uint8)
                    jpeg_host.legacy_normal_u8(seed, 0.0, float(NOISE_LEVEL), n8)
                    params["noise8"] = n8
            elif seed:   # seed 0 = "unseeded" in the reference: drawn by the parent from its global streams
                params = draw_params(op, w, h, random.Random(seed), np.random.RandomState(seed))
                if op == "distortion":
                    n8 = params.pop("noise").astype(np.uint8)   # numpy's own cast, as the reference does next
                    if n8.nbytes <= cap:
                        np.frombuffer(buf_noise, np.uint8, n8.nbytes, noff)[:] = n8.reshape(-1)
                        params["noise8"] = None      # in the noise slab, at this task's slot
                    else:
                        params["noise8"] = n8
            if arr is None:
                # a prepared scan: the shape and, fourth, how much of the slot is in use (the parent uploads only that)
                out.append(("scan", (h, w, 3, scan_hw[2]), params) if scan_hw is not None else ("coef", (h, w, 3), params))
            elif arr.nbytes <= cap:
                np.frombuffer(buf_in, np.uint8, arr.nbytes, off)[:] = arr.reshape(-1)
                out.append(("ok", arr.shape, params))
            else:
                out.append(("big", arr, params))
        except Exception as e:  # noqa: BLE001 — the reference counts any failure
            out.append(("err", f"{path} - {e}", None))
    return out


def _encode_jobs(names: Dict[str, str], jobs: Sequence[tuple]):
    """jobs: (output path, output offset, shape, inline array or None[, "px" | "coef" | "scan"]).  "scan": the
    slot holds the image's finished JPEG scan (ops.jpeg_fdct_quant_u8 + ops.jpeg_entropy_u8, quality 95) and the
    worker adds the markers; "coef": its quantised DCT coefficients, the worker Huffman-codes them
    (libleafcodec.so); otherwise pixels for Pillow.  Returns one bool per job."""
    from pathlib import Path
    from ..utils import jpeg_host
    from ..utils.image_utils import ImageLoader
    buf = _slabs(names)["out"].buf
    done = []
    for job in jobs:
        path, off, shape, inline = job[:4]
        try:
            if len(job) > 4 and job[4] in ("coef", "scan"):
                h, w = int(shape[0]), int(shape[1])
                if job[4] == "scan":   # the GPU's entropy coder has been over it: int32 length, then the scan
                    n = int(np.frombuffer(buf, np.int32, 1, off)[0])
                    if n < 0:
                        raise RuntimeError("the coded scan did not fit its slot")
                    data = jpeg_host.wrap_scan(np.frombuffer(buf, np.uint8, n, off + 4), h, w, 95)
                else:
                    data = jpeg_host.write_file(np.frombuffer(buf, np.int16, h * w * 3 // 2, off), h, w, 95)
                dst = Path(path)
                dst.parent.mkdir(parents=True, exist_ok=True)
                with open(dst, "wb") as f:
                    f.write(data)
                done.append(True)
                continue
            arr = inline if inline is not None else np.frombuffer(
                buf, np.uint8, int(np.prod(shape)), off).reshape(shape)
            ImageLoader.save_array(arr, path)
            done.append(True)
        except Exception:  # noqa: BLE001
            done.append(False)
    return done


class CodecPool:
    """`workers` codec processes (started, and warmed up, at construction) + three shared-memory
    slabs of `slots` slots of `slot_bytes` each (`allocate`, once the image size is known)."""

    def __init__(self, workers: int) -> None:
        self.workers = int(workers)
        self.slots = self.slot_bytes = 0
        self.slabs: Dict[str, shared_memory.SharedMemory] = {}
        self.names: Dict[str, str] = {}
        # spawn, not fork: the parent has an initialised HIP runtime that a forked child must not inherit
        self.pool = ProcessPoolExecutor(max_workers=self.workers, mp_context=get_context("spawn"))
        # A spawned child re-imports the parent's __main__ (bench.py, a CLI: torch and all, 1-2 s of CPU per
        # worker) before it can run anything.  Nothing these workers run lives in __main__, so the main module is
        # hidden from multiprocessing's preparation data while the processes are started.
        import sys
        main = sys.modules.get("__main__")
        saved = (getattr(main, "__file__", None), getattr(main, "__spec__", None)) if main is not None else None
        try:
            if main is not None:
                main.__file__, main.__spec__ = None, None
            self._warming = [self.pool.submit(_warm, i) for i in range(self.workers)]
        finally:
            if main is not None:
                main.__file__, main.__spec__ = saved

    def allocate(self, slots: int, slot_bytes: int) -> None:
        self.slots, self.slot_bytes = int(slots), int(slot_bytes)
        size = self.slots * self.slot_bytes
        self.slabs = {k: shared_memory.SharedMemory(create=True, size=size) for k in ("in", "noise", "out")}
        self.names = {k: s.name for k, s in self.slabs.items()}

    def pin(self) -> bool:
        """Page-lock the slabs (hipHostRegister) so that a whole chunk crosses PCIe as ONE asynchronous copy each
        way, straight from / into the memory the codec workers use (the noise slab too: a distortion task's noise
        plane goes up from its slot with an asynchronous copy of its own instead of being stacked on the host first).
        Returns False when the runtime refuses (the pageable path still works, at a third of the rate)."""
        import torch
        rt = torch.cuda.cudart()
        self._pinned = []
        for key in ("in", "out", "noise"):
            t = torch.frombuffer(self.slabs[key].buf, dtype=torch.uint8)
            if int(rt.cudaHostRegister(t.data_ptr(), t.numel(), 0)) != 0:
                return False
            self._pinned.append(t)
        return True

    def unpin(self) -> None:
        pinned = getattr(self, "_pinned", [])
        if pinned:
            import torch
            rt = torch.cuda.cudart()
            for t in pinned:
                rt.cudaHostUnregister(t.data_ptr())
        self._pinned = []

    def tensor(self, slab: str, first_slot: int, n_slots: int):
        """uint8 torch view [n_slots, slot_bytes] of a run of slots (shares the slab's memory)."""
        import torch
        t = torch.frombuffer(self.slabs[slab].buf, dtype=torch.uint8, count=n_slots * self.slot_bytes,
                             offset=first_slot * self.slot_bytes)
        return t.view(n_slots, self.slot_bytes)

    def view(self, slab: str, slot: int, shape: Tuple[int, ...]) -> np.ndarray:
        n = int(np.prod(shape))
        return np.frombuffer(self.slabs[slab].buf, np.uint8, n, slot * self.slot_bytes).reshape(shape)

    def _split(self, jobs: List[Any], pieces_per_worker: int = 4) -> List[List[Any]]:
        # four pieces per worker: a piece that happens to hold several distortion tasks (150 k normal deviates
        # each, 4x the cost of the decode next to it) no longer decides when the chunk is ready; one piece per
        # worker for plain decoding (uniform jobs: every future costs the parent ~0.1 ms to send and collect)
        per = max(1, -(-len(jobs) // (pieces_per_worker * self.workers)))
        return [jobs[i:i + per] for i in range(0, len(jobs), per)]

    def decode(self, tasks: Sequence[dict], first_slot: int, coefficients: int = 0,
               pieces_per_worker: int = 4, gpu_noise: bool = False) -> List[Future]:
        jobs = [(t.get("read_img", t["source_img"]), t["transform_name"], t["seed"], (first_slot + k) * self.slot_bytes,
                 (first_slot + k) * self.slot_bytes, self.slot_bytes, coefficients, gpu_noise) for k, t in enumerate(tasks)]
        return [self.pool.submit(_decode_jobs, self.names, part) for part in self._split(jobs, pieces_per_worker)]

    def encode(self, jobs: List[Tuple[str, int, Tuple[int, int, int], Optional[np.ndarray]]]) -> List[Future]:
        return [self.pool.submit(_encode_jobs, self.names, part) for part in self._split(jobs, 2)]

    def close(self) -> None:
        """Stop the workers and WAIT for them, then take the slabs down.

        The wait has to be a real one.  Round 2 called `shutdown(wait=False)` first and `shutdown(wait=True)` at the
        end, believing the second call joined the executor — it does not: the first call already drops the executor's
        reference to its manager thread (concurrent/futures/process.py, `self._executor_manager_thread = None`), so
        the second has nothing to join.  The manager thread was then still on its way to the workers' stop sentinels
        when a short-lived process (a rank of the two-rank job) reached its exit handlers: multiprocessing's exit
        finalizers close the call queue — its feeder thread leaves — BEFORE the sentinels are queued, the sentinels
        are never written to the pipe, the workers wait on it for ever (every worker holds a duplicate of the pipe's
        write end, so there is no EOF either), and both the manager thread and the exit handler sit in `p.join()`.
        Those are exactly the stacks a stuck rank and its workers left (tests/test_sharded_paths.py, round 3: ranks in
        `join_executor_internals -> p.join()` and `_exit_function -> p.join()`, workers in `call_queue.get()`)."""
        self.pool.shutdown(wait=True, cancel_futures=True)   # every result has been collected; returns once the workers are gone
        self.unpin()
        import gc
        gc.collect()   # numpy views of the slabs that are only kept alive by cycles
        for s in self.slabs.values():
            try:
                s.unlink()          # the name goes now; the pages when the last mapping does
            except FileNotFoundError:
                pass
            try:
                s.close()
            except BufferError:     # a caller still holds a view: its mapping outlives the pool, harmlessly
                pass
