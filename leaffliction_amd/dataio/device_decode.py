"""Files -> resized uint8 batches in HBM: the decode step either side of every kernel.

The reference decodes one file after the other with Pillow and resizes on the host
(`ImageLoader.load_as_array` / `resize_array`, srcs/utils/image_utils.py:19-59,109-114, called from
srcs/dataio/sequence.py:74-125 and srcs/predict/predictor.py).  Here codec worker processes read the
files and, of baseline 4:2:0 JPEGs of whole MCUs, either the markers only (`chunks`: the un-stuffed scan goes into the
page-locked slab and the GPU decodes the Huffman stream too, `ops.jpeg_huffman_u8`) or the markers and the Huffman
stream (the prefetching `submit` / `collect`, which never waits for the GPU and so cannot ask it for a verdict on a
damaged scan); any other file is decoded whole by Pillow in the worker.  A chunk crosses PCIe as one copy (its used
parts only when every file is a prepared scan), and the GPU does dequantisation / IDCT / fancy upsampling / colour
conversion (`ops.jpeg_idct_rgb_u8`) and the Pillow-exact
LANCZOS resize, chunk after chunk with the next two chunks' files already being read.  The pixels are
Pillow's bit for bit (tests/test_jpeg_codec.py), so everything downstream sees what the reference's loop
would have produced.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np


class DeviceDecoder:
    """`chunks(paths, img_size)` yields `(first, kept, x, natives, errors)` per chunk of up to `CHUNK` files:
    `kept` = positions (into `paths`) that decoded, in order; `x` = uint8 device tensor [len(kept), S, S, 3];
    `natives` = the decoded images at their own size (host arrays, `keep_native=True` only, else None);
    `errors` = [(position, message)] for the files that did not decode.  The worker processes and their
    slabs live until `close()`: starting them costs as much as decoding a thousand files.
    `submit(paths, img_size)` / `collect(handle)` split one batch into its worker half and its device half, so a
    loader can have the next batch's files decoding while the current one trains."""

    CHUNK = 256

    def __init__(self, workers: Optional[int] = None) -> None:
        from ..utils.system_info import get_optimal_worker_count
        self.workers = int(workers or get_optimal_worker_count())   # (already the rank's share of the node's cores)
        self._codec = None   # (CodecPool, slot_bytes, pinned, device staging)
        self._pending: List[dict] = []   # submitted, not yet collected batches (at most two)
        self._uploaded = [None, None, None]   # per slab third: event behind its last (asynchronous) upload
        self._ring = 0                   # next slab third `submit` uses

    def _ensure(self, slot: int):
        import torch

        from ..preprocessing.codec_pool import CodecPool
        if self._codec is not None and self._codec[1] < slot:
            self.close()
        if self._codec is None:
            pool = CodecPool(self.workers)
            pool.allocate(3 * self.CHUNK, slot)
            dev = torch.device("cuda", torch.cuda.current_device())
            self._codec = (pool, slot, pool.pin(), torch.empty((self.CHUNK, slot), dtype=torch.uint8, device=dev))
        return self._codec

    @staticmethod
    def _probe_slot(paths: Sequence[str], fallback: int, scan: bool = False) -> int:
        from PIL import Image
        w0 = h0 = fallback
        for p in paths[:8]:   # slot size from the first readable file (larger images travel as pickled arrays)
            try:
                with Image.open(p) as probe:
                    w0, h0 = probe.size
                break
            except Exception:  # noqa: BLE001
                continue
        # tables + coefficients (or pixels), and behind them room for a prepared scan (header, Huffman tables and the
        # un-stuffed entropy-coded bytes: ~0.16 of the pixel bytes at quality 95; a file that does not fit takes the
        # host's Huffman pass)
        return (256 + h0 * w0 * 3 + ((1136 + h0 * w0 * 3 // 2) if scan else 0) + 4095) // 4096 * 4096

    def _submit(self, part: Sequence[str], third: int, scan: bool = False):
        """`scan`: the workers read the markers only and the GPU decodes the Huffman stream as well — for callers that
        can wait for the chunk's GPU work before they use its pixels (`chunks`); the prefetching path (`submit` /
        `collect`, which never waits) keeps the Huffman pass in the workers."""
        pool = self._codec[0]
        ev = self._uploaded[third]   # the last upload out of this slab third must be over before it is rewritten
        if ev is not None:
            ev.synchronize()
            self._uploaded[third] = None
        tasks = [{"source_img": p, "transform_name": "", "seed": 0} for p in part]
        # every future costs the parent ~0.1-0.2 ms to send and collect: one job per worker for a small batch (a
        # 32-file batch as 32 one-file jobs ran at half the rate), two for a full chunk (some slack for a slow core)
        return pool.decode(tasks, third * self.CHUNK, 2 if scan else 1, pieces_per_worker=1 if len(part) <= 64 else 2)

    def _finish(self, futures, third: int, n: int, img_size: int, keep_native: bool, pos0: int,
                paths: Optional[Sequence[str]] = None):
        """The device half of one chunk whose worker jobs are `futures`: upload, the JPEG back end, the resize.
        Returns (kept positions, x, natives, errors); the slab third is free again once `_uploaded[third]` has passed."""
        import torch

        from .. import ops
        pool, _slot, pinned, dev_in = self._codec
        S, dev = int(img_size), dev_in.device
        decoded = [r for f in futures for r in f.result()]
        host = pool.tensor("in", third * self.CHUNK, n)
        # Nothing below waits for the GPU (unless `keep_native` fetches pixels back): the device half queues
        # behind whatever the stream is doing — a training step — and the host goes on to prepare the next one.
        # The one staging buffer is safe to reuse: uploads and the kernels that read it are ordered on the stream.
        dev_in = dev_in[:n]
        live = [d for d in decoded if d[0] != "err"]
        if pinned and live and all(d[0] == "scan" for d in live):
            # prepared scans only: the tables at the front of each slot and the scan behind the (device-only)
            # coefficient area are all that has to cross PCIe
            from .. import _lib
            from ..utils import jpeg_host
            lo = min(jpeg_host.scan_aux_offset(hh, ww) for hh, ww in {(d[1][0], d[1][1]) for d in live})
            hi = min(_slot, (max(d[1][3] for d in live) + 15) // 16 * 16)
            stream = torch.cuda.current_stream().cuda_stream
            _lib.call("lf_copy_rows", dev_in.data_ptr(), _slot, host.data_ptr(), _slot, 256, n, 0, stream)
            _lib.call("lf_copy_rows", dev_in.data_ptr() + lo, _slot, host.data_ptr() + lo, _slot, hi - lo, n, 0, stream)
        else:
            dev_in.copy_(host if pinned else host.clone(), non_blocking=pinned)
        if pinned:
            ev = torch.cuda.Event()
            ev.record()
            self._uploaded[third] = ev

        def index(ks):
            t = torch.tensor(ks, dtype=torch.int64)
            return (t.pin_memory() if pinned else t).to(dev, non_blocking=pinned)
        groups: Dict[tuple, List[int]] = {}
        big: Dict[int, np.ndarray] = {}
        errors: List[Tuple[int, str]] = []
        for k, (status, payload, _prm) in enumerate(decoded):
            if status == "err":
                errors.append((pos0 + k, payload))
            elif status == "big":
                big[k] = payload
                groups.setdefault(("big",) + tuple(payload.shape[:2]), []).append(k)
            else:
                groups.setdefault((status,) + tuple(payload[:2]), []).append(k)
        natives: Optional[Dict[int, np.ndarray]] = {} if keep_native else None
        x = torch.empty((n, S, S, 3), dtype=torch.uint8, device=dev)
        huffman: List[tuple] = []   # (positions in the chunk, device status of the GPU's Huffman decoding)
        for (status, h, w), ks in groups.items():
            whole = len(ks) == n   # one group holds the whole chunk (the usual case): no gather, no scatter
            idx = None if whole else index(ks)
            if status == "scan":
                rows = dev_in if whole else dev_in[idx]
                huffman.append((ks, ops.jpeg_huffman_u8(rows, h, w)))
                px = ops.jpeg_idct_rgb_u8(rows, h, w)
            elif status == "coef":
                px = ops.jpeg_idct_rgb_u8(dev_in if whole else dev_in[idx], h, w)
            elif status == "ok":
                px = (dev_in if whole else dev_in[idx])[:, :h * w * 3].view(len(ks), h, w, 3)
            else:
                px = torch.from_numpy(np.stack([big[k] for k in ks])).to(dev)
            res = px if (h, w) == (S, S) else ops.resize_lanczos_u8(px.contiguous(), S)
            if whole:   # (a view of the staging buffer is never contiguous: it gets copied here)
                x = res if res.is_contiguous() else res.contiguous()
            else:
                x[idx] = res
            if natives is not None:
                host_px = px.cpu().numpy()
                for j, k in enumerate(ks):
                    natives[pos0 + k] = host_px[j]
        kept = sorted(k for ks in groups.values() for k in ks)
        for ks, st in huffman:
            st = st.cpu().numpy()   # waits for the chunk's device half: only `chunks` asks for prepared scans
            for k in (k for k, v in zip(ks, st) if v):
                # a scan the GPU handed back (damaged, cut short): Pillow has the reference's verdict
                # (image_utils.py:19-33) — pixels, with libjpeg's concealment, or an error
                try:
                    from ..utils.image_utils import ImageLoader
                    arr = ImageLoader.load_as_array(paths[k])
                    one = torch.from_numpy(np.ascontiguousarray(arr)).to(dev).unsqueeze(0)
                    x[k] = (one if tuple(arr.shape[:2]) == (S, S) else ops.resize_lanczos_u8(one, S))[0]
                    if natives is not None:
                        natives[pos0 + k] = arr
                except Exception as e:  # noqa: BLE001
                    errors.append((pos0 + k, f"{paths[k]} - {e}"))
                    kept.remove(k)
                    if natives is not None:
                        natives.pop(pos0 + k, None)
        if len(kept) < n:
            x = x[index(kept)] if kept else x[:0]
        return [pos0 + k for k in kept], x, natives, errors

    def drop_pending(self) -> None:
        """Wait for (and drop) what `submit` left pending — the synchronous path is about to use every third, or
        the caller no longer wants those batches; `collect` on a dropped handle returns None."""
        for h in self._pending:
            h["dropped"] = True
            for f in h["futures"]:
                try:
                    f.result()
                except Exception:  # noqa: BLE001
                    pass
        self._pending = []
        self._ring = 0

    def chunks(self, paths: Sequence, img_size: int, keep_native: bool = False) -> Iterator[
            Tuple[int, List[int], "object", Optional[Dict[int, np.ndarray]], List[Tuple[int, str]]]]:
        C = self.CHUNK
        paths = [str(p) for p in paths]
        self.drop_pending()
        self._ensure(self._probe_slot(paths, int(img_size), scan=True))
        parts = [paths[b:b + C] for b in range(0, len(paths), C)]
        try:
            ahead = [self._submit(parts[i], i % 3, scan=True) for i in range(min(2, len(parts)))]
            for i, part in enumerate(parts):
                futures = ahead.pop(0)
                for f in futures:
                    f.result()
                if i + 2 < len(parts):   # into the slab third chunk i-1 used: its upload was waited for
                    ahead.append(self._submit(parts[i + 2], (i + 2) % 3, scan=True))
                kept, x, natives, errors = self._finish(futures, i % 3, len(part), img_size, keep_native, i * C, part)
                yield i * C, kept, x, natives, errors
        except BaseException:
            self.close()   # a failed chunk may leave jobs in flight on the slabs: start afresh next time
            raise

    # ---- one batch ahead: the worker half now, the device half when the batch is asked for
    def submit(self, paths: Sequence, img_size: int):
        """Start the worker half (file read + Huffman decoding) of one batch of at most CHUNK files and return a
        handle for `collect`, or None when two batches are already pending (the slab ring has three thirds)."""
        paths = [str(p) for p in paths]
        if not paths or len(paths) > self.CHUNK or len(self._pending) >= 2:
            return None
        slot = self._probe_slot(paths, int(img_size))
        if self._codec is None or self._codec[1] < slot:
            self.drop_pending()
            self._ensure(slot)
        third = self._ring % 3
        self._ring += 1
        h = {"futures": self._submit(paths, third), "third": third, "n": len(paths), "S": int(img_size)}
        self._pending.append(h)
        return h

    def collect(self, h):
        """The device half of a submitted batch: (kept positions, uint8 device tensor [len(kept), S, S, 3], errors);
        None when the handle was dropped in the meantime."""
        if h.get("dropped") or not any(q is h for q in self._pending):
            return None
        try:
            kept, x, _nat, errors = self._finish(h["futures"], h["third"], h["n"], h["S"], False, 0)
        except BaseException:
            self.close()
            raise
        self._pending = [q for q in self._pending if q is not h]
        return kept, x, errors

    def close(self) -> None:
        """Stop the codec workers and release their slabs (idempotent)."""
        codec, self._codec = self._codec, None
        self._pending, self._ring = [], 0
        if codec is not None:
            import torch
            if any(ev is not None for ev in self._uploaded):
                torch.cuda.synchronize()   # uploads out of the slabs that are about to be unmapped
            self._uploaded = [None, None, None]
            codec[0].close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass
