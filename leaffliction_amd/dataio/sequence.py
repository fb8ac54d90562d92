"""Batch loader with the reference's `ManifestSequence` contract (srcs/dataio/sequence.py:16-176).

Same constructor, same batching rules (ceil(len/batch) batches, short last batch, seeded
`random.Random(seed)` shuffle at construction and on every `on_epoch_end`), same label
formats (one-hot float32 [C] or int32 index), same optional `transform` hook
`(Path, ManifestItem, img_size) -> (uint8 HxWx3, float32 HxWx3)` and RAM cache.

Differences that matter on MI355X: images are decoded on host threads (Pillow) but resized
(Pillow-exact LANCZOS kernel) and packed on the GPU; `__getitem__` returns the batch as a
uint8 device tensor [B,S,S,3] by default (`as_numpy=True` gives the reference's float32
NHWC host array instead).  For data-parallel training pass `rank`/`world`: every rank keeps
the same permutation and takes the rank-strided slice of each global batch.  With
`cache=True` (and device batches) the resized uint8 dataset lives in HBM — 150 KB per 224x224
image, so even 100 k images are 15 GB of the 288 GB — and a batch is one gather kernel; there
is no per-step decode, host stack or PCIe upload.  Device batches of `POOL_MIN` files or more
(and the filling of that cache) go through `dataio/device_decode.py`: codec worker processes
Huffman-decode the JPEGs, the GPU finishes the decoding and resizes — Pillow's pixels, bit for bit.
"""
from __future__ import annotations

import math
import random
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np

from .manifest import ManifestItem
from ..utils.image_utils import ImageLoader


class ManifestSequence:
    def __init__(self, items: List[ManifestItem], label2idx: Optional[Dict[str, int]], img_size: int,
                 batch_size: int, shuffle: bool, seed: int, limit: Optional[int] = None,
                 num_classes: Optional[int] = None, one_hot: bool = False, cache: bool = False,
                 workers: int = 1,
                 transform: Optional[Callable[[Path, ManifestItem, int],
                                              Tuple[np.ndarray, np.ndarray]]] = None,
                 as_numpy: bool = False, rank: int = 0, world: int = 1, **kwargs) -> None:
        if limit is not None:
            items = items[:limit]
        self.items = items
        self.label2idx = label2idx
        self.img_size = img_size
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.rng = random.Random(seed)
        self.indexes = list(range(len(items)))
        self.one_hot = one_hot and (label2idx is not None)
        if self.one_hot and num_classes is None:
            raise ValueError("num_classes must be provided when one_hot=True")
        self.num_classes = int(num_classes or 0)
        self.cache = cache
        self.workers = max(1, int(workers))
        self.transform = transform
        self.as_numpy = as_numpy
        self.rank, self.world = int(rank), max(1, int(world))
        self._cache_u8: Dict[int, np.ndarray] = {}
        self._cache_dev = None  # uint8 device tensor [N,S,S,3] (cache=True, device batches)
        self._decoder = None    # DeviceDecoder (device batches of POOL_MIN files or more)
        self._ahead: Dict[int, tuple] = {}   # batch index -> (decoder handle, item indexes) started by prefetch()
        if self.shuffle:
            self.rng.shuffle(self.indexes)
        if self.cache:
            self._build_cache()

    def __len__(self) -> int:
        return math.ceil(len(self.items) / self.batch_size)

    def on_epoch_end(self) -> None:
        if self.shuffle:
            self.rng.shuffle(self.indexes)
        if self._ahead:       # started for the old order
            self._ahead.clear()
            if self._decoder is not None:
                self._decoder.drop_pending()

    # ------------------------------------------------------------------ loading
    def _decode(self, i: int) -> np.ndarray:
        """Decoded RGB uint8 at native size (host)."""
        return ImageLoader.load_as_array(self.items[i].src)

    def _label(self, i: int):
        if self.label2idx is None:
            return None
        idx = self.label2idx[self.items[i].label]
        if self.one_hot:
            la = np.zeros(self.num_classes, dtype="float32")
            la[idx] = 1.0
            return la
        return np.asarray(idx, dtype="int32")

    def _resize_group(self, arrays: List[np.ndarray]) -> List[np.ndarray]:
        """LANCZOS-resize decoded images to img_size on the GPU, grouped by native size."""
        import torch

        from .. import ops
        S = self.img_size
        out: List[Optional[np.ndarray]] = [None] * len(arrays)
        groups: Dict[Tuple[int, int], List[int]] = {}
        for k, a in enumerate(arrays):
            groups.setdefault(a.shape[:2], []).append(k)
        for (h, w), ks in groups.items():
            if (h, w) == (S, S):
                for k in ks:
                    out[k] = arrays[k]
                continue
            batch = torch.from_numpy(np.stack([arrays[k] for k in ks])).cuda()
            res = ops.resize_lanczos_u8(batch, S).cpu().numpy()
            for j, k in enumerate(ks):
                out[k] = res[j]
        return out  # type: ignore[return-value]

    def _load_u8(self, idxs: List[int]) -> np.ndarray:
        """uint8 [B,S,S,3] for the given item indexes (cache-aware)."""
        missing = [i for i in idxs if i not in self._cache_u8]
        if missing:
            if self.transform is not None:
                loaded = [self.transform(Path(self.items[i].src), self.items[i], self.img_size)[0]
                          for i in missing]
            else:
                if self.workers > 1:
                    with ThreadPoolExecutor(max_workers=self.workers) as ex:
                        decoded = list(ex.map(self._decode, missing))
                else:
                    decoded = [self._decode(i) for i in missing]
                loaded = self._resize_group(decoded)
            fresh = dict(zip(missing, loaded))
            if self.cache:
                self._cache_u8.update(fresh)
        else:
            fresh = {}
        return np.stack([self._cache_u8[i] if i in self._cache_u8 else fresh[i] for i in idxs])

    POOL_MIN = 64   # below this many files the worker pool costs more than it saves

    def _load_dev(self, idxs: List[int]):
        """uint8 device tensor [B,S,S,3] for the given item indexes: pooled decode + GPU JPEG back end
        when the batch is large enough, else the host path and one upload.  An unreadable file raises,
        as the reference's loader does (image_utils.py:19-33)."""
        import torch
        if self.transform is not None or len(idxs) < self.POOL_MIN or any(i in self._cache_u8 for i in idxs):
            return torch.from_numpy(self._load_u8(idxs)).cuda()
        from .device_decode import DeviceDecoder
        if self._decoder is None:
            self._decoder = DeviceDecoder(self.workers if self.workers > 1 else None)
        parts = []
        for _first, _kept, x, _nat, errors in self._decoder.chunks([self.items[i].src for i in idxs], self.img_size):
            if errors:
                raise OSError(f"Cannot load image {errors[0][1]}")
            parts.append(x)
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    PREFETCH_MIN = 16   # the worker pool is already up: worth it from a few files on

    def prefetch(self, idx: int) -> None:
        """Start decoding batch `idx` on the codec workers (file reads + Huffman decoding; nothing on the GPU yet)
        so that `self[idx]` only has the device half left.  `fit` calls it for the next batch before each step;
        a no-op for host batches, cached datasets, transforms, and batches too small or too large for one chunk."""
        if (self.as_numpy or self.cache or self._cache_dev is not None or self.transform is not None
                or idx < 0 or idx >= len(self) or idx in self._ahead):
            return
        import torch
        if not torch.cuda.is_available():
            return
        batch_idx = self.batch_indexes(idx)
        from .device_decode import DeviceDecoder
        if not self.PREFETCH_MIN <= len(batch_idx) <= DeviceDecoder.CHUNK:
            return
        if self._decoder is None:
            self._decoder = DeviceDecoder(self.workers if self.workers > 1 else None)
        h = self._decoder.submit([self.items[i].src for i in batch_idx], self.img_size)
        if h is not None:
            self._ahead[idx] = (h, batch_idx)

    def close(self) -> None:
        """Stop the decoder's worker processes (idempotent; also done when the sequence is collected)."""
        self._ahead.clear()
        dec, self._decoder = self._decoder, None
        if dec is not None:
            dec.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass

    def _build_cache(self) -> None:
        step = max(64, self.batch_size)
        dev_ok = not self.as_numpy and len(self.items) > 0
        if dev_ok:
            import torch
            dev_ok = torch.cuda.is_available()
        if not dev_ok:
            for b in range(0, len(self.items), step):
                self._load_u8(list(range(b, min(b + step, len(self.items)))))
            return
        # device-resident dataset: fill chunk by chunk, nothing is kept on the host
        import torch
        S = self.img_size
        self._cache_dev = torch.empty((len(self.items), S, S, 3), dtype=torch.uint8, device="cuda")
        keep, self.cache = self.cache, False  # _load_u8 must not also fill the host cache
        try:
            step = max(step, 4096)
            for b in range(0, len(self.items), step):
                e = min(b + step, len(self.items))
                self._cache_dev[b:e].copy_(self._load_dev(list(range(b, e))))
        finally:
            self.cache = keep
            self.close()   # the cache is complete: nothing is decoded after this

    def batch_indexes(self, idx: int) -> List[int]:
        start = idx * self.batch_size
        end = min(start + self.batch_size, len(self.items))
        return self.indexes[start:end][self.rank::self.world]

    def global_batch_size(self, idx: int) -> int:
        start = idx * self.batch_size
        return min(start + self.batch_size, len(self.items)) - start

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __getitem__(self, idx: int):
        if idx < 0 or idx >= len(self):
            raise IndexError(idx)
        batch_idx = self.batch_indexes(idx)
        if self._cache_dev is not None and batch_idx and not self.as_numpy:
            import torch

            from .. import ops
            sel = torch.tensor(batch_idx, dtype=torch.int32).pin_memory().cuda(non_blocking=True)
            X = ops.gather_images_u8(self._cache_dev, sel)
            if self.label2idx is None:
                return X
            return X, np.asarray([self._label(i) for i in batch_idx])
        if self.as_numpy or not batch_idx:
            x_u8 = self._load_u8(batch_idx) if batch_idx else np.zeros(
                (0, self.img_size, self.img_size, 3), np.uint8)
            if not self.as_numpy:
                import torch
                x_u8 = torch.from_numpy(x_u8).cuda()
        else:
            ahead = self._ahead.pop(idx, None)
            got = None
            if ahead is not None and ahead[1] == batch_idx and self._decoder is not None:
                got = self._decoder.collect(ahead[0])
            if got is not None:
                _kept, x_u8, errors = got
                if errors:
                    raise OSError(f"Cannot load image {errors[0][1]}")
            else:
                x_u8 = self._load_dev(batch_idx)
        if self.as_numpy:
            X = x_u8.astype(np.float32) / 255.0  # normalize_array (image_utils.py:117-130)
        else:
            X = x_u8
        if self.label2idx is None:
            return X
        y = np.asarray([self._label(i) for i in batch_idx])
        return X, y

    def iter_with_info(self, batch_size: Optional[int] = None) -> Iterable[
            Tuple[np.ndarray, Optional[np.ndarray], List[np.ndarray], List[ManifestItem]]]:
        """Yield (X_float32 NHWC, y_or_None, originals_uint8_list, items_list) like the reference."""
        bs = batch_size or self.batch_size
        total = len(self.items)
        for start in range(0, total, bs):
            idxs = self.indexes[start:min(start + bs, total)]
            x_u8 = self._load_u8(idxs)
            yb = None
            if self.label2idx is not None:
                yb = np.asarray([self._label(i) for i in idxs])
            yield (x_u8.astype(np.float32) / 255.0, yb, [x_u8[k] for k in range(len(idxs))],
                   [self.items[i] for i in idxs])
