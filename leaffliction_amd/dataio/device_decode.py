"""Files -> resized uint8 batches in HBM: the decode step either side of every kernel.

The reference decodes one file after the other with Pillow and resizes on the host
(`ImageLoader.load_as_array` / `resize_array`, srcs/utils/image_utils.py:19-59,109-114, called from
srcs/dataio/sequence.py:74-125 and srcs/predict/predictor.py).  Here codec worker processes read the
files and Huffman-decode baseline 4:2:0 JPEGs of whole MCUs into page-locked slabs (libleafcodec.so; any
other file is decoded whole by Pillow in the worker), a chunk crosses PCIe as one copy, and the GPU does
dequantisation / IDCT / fancy upsampling / colour conversion (`ops.jpeg_idct_rgb_u8`) and the Pillow-exact
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
    slabs live until `close()`: starting them costs as much as decoding a thousand files."""

    CHUNK = 256

    def __init__(self, workers: Optional[int] = None) -> None:
        from ..utils.system_info import get_optimal_worker_count
        self.workers = int(workers or get_optimal_worker_count())
        self._codec = None   # (CodecPool, slot_bytes, pinned, device staging)

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
    def _probe_slot(paths: Sequence[str], fallback: int) -> int:
        from PIL import Image
        w0 = h0 = fallback
        for p in paths[:8]:   # slot size from the first readable file (larger images travel as pickled arrays)
            try:
                with Image.open(p) as probe:
                    w0, h0 = probe.size
                break
            except Exception:  # noqa: BLE001
                continue
        return (256 + h0 * w0 * 3 + 4095) // 4096 * 4096

    def chunks(self, paths: Sequence, img_size: int, keep_native: bool = False) -> Iterator[
            Tuple[int, List[int], "object", Optional[Dict[int, np.ndarray]], List[Tuple[int, str]]]]:
        import torch

        from .. import ops
        S, C = int(img_size), self.CHUNK
        paths = [str(p) for p in paths]
        pool, slot, pinned, dev_in = self._ensure(self._probe_slot(paths, S))
        dev = dev_in.device
        parts = [paths[b:b + C] for b in range(0, len(paths), C)]

        def submit(i):
            tasks = [{"source_img": p, "transform_name": "", "seed": 0} for p in parts[i]]
            return pool.decode(tasks, (i % 3) * C, True)

        try:
            ahead = [submit(i) for i in range(min(2, len(parts)))]
            for i, part in enumerate(parts):
                decoded = [r for f in ahead.pop(0) for r in f.result()]
                if i + 2 < len(parts):   # into the slab third chunk i-1 used: its upload was waited for below
                    ahead.append(submit(i + 2))
                n, base = len(part), (i % 3) * C
                host = pool.tensor("in", base, n)
                dev_in[:n].copy_(host if pinned else host.clone(), non_blocking=pinned)
                groups: Dict[tuple, List[int]] = {}
                big: Dict[int, np.ndarray] = {}
                errors: List[Tuple[int, str]] = []
                for k, (status, payload, _prm) in enumerate(decoded):
                    if status == "err":
                        errors.append((i * C + k, payload))
                    elif status == "big":
                        big[k] = payload
                        groups.setdefault(("big",) + tuple(payload.shape[:2]), []).append(k)
                    else:
                        groups.setdefault((status,) + tuple(payload[:2]), []).append(k)
                natives: Optional[Dict[int, np.ndarray]] = {} if keep_native else None
                x = torch.empty((n, S, S, 3), dtype=torch.uint8, device=dev)
                for (status, h, w), ks in groups.items():
                    idx = torch.tensor(ks, dtype=torch.int64, device=dev)
                    if status == "coef":
                        px = ops.jpeg_idct_rgb_u8(dev_in[idx], h, w)
                    elif status == "ok":
                        px = dev_in[idx, :h * w * 3].view(len(ks), h, w, 3)
                    else:
                        px = torch.from_numpy(np.stack([big[k] for k in ks])).to(dev)
                    x[idx] = px if (h, w) == (S, S) else ops.resize_lanczos_u8(px.contiguous(), S)
                    if natives is not None:
                        host_px = px.cpu().numpy()
                        for j, k in enumerate(ks):
                            natives[i * C + k] = host_px[j]
                kept = sorted(k for ks in groups.values() for k in ks)
                if len(kept) < n:
                    x = x[torch.tensor(kept, dtype=torch.int64, device=dev)] if kept else x[:0]
                # this chunk's slab third is handed to the workers again at the next iteration: its
                # (asynchronous) upload must have finished by then
                torch.cuda.current_stream().synchronize()
                yield i * C, [i * C + k for k in kept], x, natives, errors
        except BaseException:
            decoded = host = None
            self.close()   # a failed chunk may leave jobs in flight on the slabs: start afresh next time
            raise

    def close(self) -> None:
        """Stop the codec workers and release their slabs (idempotent)."""
        codec, self._codec = self._codec, None
        if codec is not None:
            codec[0].close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass
