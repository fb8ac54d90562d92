"""Dataset balancing on MI355X behind the reference's `DatasetBalancer` contract
(srcs/preprocessing/dataset_balancer.py:19-207): `DatasetBalancer(manifest_path, source_dir,
target_dir, seed, workers).run()` leaves the balanced tree and `manifest_augmented.json`.

What is kept from the reference, because outputs depend on it: the task list.  Plan order,
`random.choice(source_images)` then `random.randint(0, 10**6)` per task on the process-global
stream, and the `<stem>_aug_<transform>_<k><suffix>` names (dataset_balancer.py:105-129) —
so sources, names and per-task seeds are the reference's.

What is different: execution.  The reference hands one PIL call per task to a process pool
(:137-141).  Here the list is cut into one contiguous share per GPU (rank-0 builds it and
broadcasts it) and every share runs as a three-stage pipeline over chunks —
JPEG decode in worker processes, one batched kernel launch per (transform, image size) group,
JPEG encode in worker processes (pixels cross through shared memory: codec_pool.py) — with the
stages of neighbouring chunks overlapping.  There is no
exchange step: ranks meet at a barrier, their success/failure counts are summed, and rank 0
writes the manifest.  Pixels are bit-identical to the reference's for every seeded task.
"""
from __future__ import annotations

import os
import random
import shutil
import time
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from .codec_pool import CodecPool
from .dataset_components import AugmentationPlanner, DistributionAnalyzer, ManifestGenerator
from .image_augmenter import ImageAugmenter, apply_batch, draw_params
from ..utils.common import get_logger
from ..utils import ranks as _ranks
from ..utils.ranks import contiguous_share
from ..utils.system_info import get_available_cores, get_optimal_worker_count

logger = get_logger(__name__)

CHUNK = int(os.environ.get("LEAFFLICTION_CHUNK", "256"))  # tasks per GPU round
RING = 4    # chunks of slots: two decoding ahead, one on the GPU, one encoding
JPEG_QUALITY = 95  # ImageLoader.save_pil_image's default (srcs/utils/image_utils.py:50)


class DatasetBalancer:
    def __init__(self, manifest_path=None, source_dir="images", target_dir="augmented_directory",
                 seed=42, workers=None):
        self.manifest_path = Path(manifest_path) if manifest_path else None
        self.source_dir, self.target_dir = Path(source_dir), Path(target_dir)
        # like the reference, constructing the augmenter is what seeds the global RNG streams
        # the task list is drawn from (image_augmenter.py:15-18)
        self.transformer = ImageAugmenter(seed=seed)
        self.workers = self._host_threads(workers)
        self.analyzer = DistributionAnalyzer(self.source_dir)
        self.planner: Optional[AugmentationPlanner] = None
        self.manifest_generator: Optional[ManifestGenerator] = None
        self.plan: Dict = {}
        self.tasks: List[dict] = []
        self.completed = self.failed = 0
        self.ranks = _ranks.current()
        self.timings: Dict[str, float] = {}   # seconds per stage of the last execute_balancing()
        self._mirror = None                   # device mirrors of one chunk of the input / output slabs
        self._codec: Optional[CodecPool] = None

    @staticmethod
    def _host_threads(requested) -> int:
        """`--workers` keeps the reference's meaning (default = half the optimal count;
        dataset_balancer.py:41-57); here they are the JPEG codec PROCESSES.  An explicit request is honoured up
        to the cores the process may use (the reference caps it at the optimal count, 3/4 of them: its workers
        compete with a busy parent, these with one that mostly waits for the GPU)."""
        ceiling = get_optimal_worker_count() if requested is None else get_available_cores()
        n = max(1, ceiling // 2) if requested is None else max(1, int(requested))
        if n > ceiling:
            logger.warning(f"Requested {n} workers, but only {ceiling} CPUs available; using {ceiling}")
            n = ceiling
        logger.info(f"Using {n} codec worker processes for JPEG decode/encode (max available: {ceiling})")
        return n

    # ------------------------------------------------------------------ planning (host)
    def analyze_distribution(self):
        counts = self.analyzer.analyze()
        self.analyzer.display_distribution()
        return counts

    def calculate_plan(self):
        self.planner = AugmentationPlanner(self.analyzer.counts)
        self.plan = self.planner.calculate_plan()
        return self.plan

    def _fresh_target(self) -> None:
        """The balanced tree starts as a copy of the originals (dataset_balancer.py:70-81)."""
        if not self.source_dir.exists():
            raise FileNotFoundError(f"Source directory not found: {self.source_dir}")
        logger.info(f"Preparing target directory: {self.target_dir}")
        if self.target_dir.exists():
            shutil.rmtree(self.target_dir)
        # The tree is laid out first (directories and EMPTY files, in copytree's own order — scandir order, a
        # directory recursed into where it comes — so that the target's directory order, the order `random.choice`
        # indexes into, is the one a plain copy gives), the bytes follow on a thread while the workers already
        # decode from the source tree (`read_img`): the copy of the originals was a quarter of the whole job.
        # execute_balancing joins the thread before the manifest.  The placeholders of DIFFERENT leaf directories
        # are created side by side (the syscalls release the GIL): on the GPU box's overlay file system one creation
        # costs 70-120 us, 8,500 of them a third of the whole job when done one after the other.
        per_dir: List[List[tuple]] = []   # leaf directories: their files, in order, laid out in parallel
        pending: List[tuple] = []
        dirs: List[tuple] = []

        def touch(dst: str) -> None:
            os.close(os.open(dst, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666))

        def walk(src: str, dst: str) -> None:
            os.makedirs(dst)
            dirs.append((src, dst))
            with os.scandir(src) as it:
                entries = list(it)
            mixed = any(e.is_dir() for e in entries)   # files next to directories: strictly in order, here
            files: List[tuple] = []
            for e in entries:
                d = dst + os.sep + e.name
                if e.is_dir():
                    walk(e.path, d)
                elif mixed:
                    touch(d)
                    pending.append((e.path, d))
                else:
                    files.append((e.path, d))
            if files:
                per_dir.append(files)

        walk(str(self.source_dir), str(self.target_dir))

        def lay(files: List[tuple]) -> None:
            for _src, dst in files:
                touch(dst)

        laid = False
        starting = getattr(self, "_codec", None)
        if starting is not None and len(per_dir) > 1:
            # one job per leaf directory for the codec workers that are starting up anyway: processes create files side
            # by side, threads of this one mostly take turns at the interpreter lock
            from .codec_pool import touch_files
            try:
                # (the largest directories first: they are what the layout waits for)
                futs = [starting.pool.submit(touch_files, [dst for _src, dst in files])
                        for files in sorted(per_dir, key=len, reverse=True)]
                for f in futs:
                    f.result()
                laid = True
            except OSError:
                raise
            except Exception:  # noqa: BLE001 — no usable pool: the threads below
                laid = False
        if not laid:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=8) as ex:
                list(ex.map(lay, per_dir))
        for src, dst in reversed(dirs):   # copytree copies a directory's stat once its entries are in place
            shutil.copystat(src, dst)
        pending += [pair for files in per_dir for pair in files]

        self._copy_error = None
        self._copy_boost = False
        pool = getattr(self, "_codec", None)

        def fill():
            # The bytes are copied by the codec worker processes, a batch of files per job, a few jobs in flight (as
            # many as there are workers once the pipeline has ended): the workers have time to spare since the Huffman
            # decoding and the noise planes went to the GPU, and a copying thread in THIS process took the interpreter
            # lock from the thread that drives the GPU stage.  Whatever the pool does not take (it was closed, it
            # broke, there is none) is copied here.
            from .codec_pool import copy_files
            try:
                batches = [pending[b:b + 64] for b in range(0, len(pending), 64)]
                flying: List[tuple] = []
                nxt = 0
                while nxt < len(batches) or flying:
                    window = 1
                    if pool is not None:
                        window = pool.workers if self._copy_boost else max(1, pool.workers // 4)
                    while nxt < len(batches) and len(flying) < window:
                        fut = None
                        if pool is not None:
                            try:
                                fut = pool.pool.submit(copy_files, batches[nxt])
                            except Exception:  # noqa: BLE001 — shut down or broken
                                fut = None
                        if fut is None:
                            copy_files(batches[nxt])
                        else:
                            flying.append((fut, nxt))
                        nxt += 1
                    if flying:
                        fut, b = flying.pop(0)
                        try:
                            fut.result()
                        except (OSError, shutil.Error):
                            raise
                        except BaseException:  # noqa: BLE001 — cancelled by the pool's shutdown, or the pool broke
                            copy_files(batches[b])
            except BaseException as e:  # noqa: BLE001 — re-raised on the main thread by _join_copy
                self._copy_error = e

        import threading
        self._copying = threading.Thread(target=fill, name="copy-originals", daemon=True)
        self._copying.start()

    def _join_copy(self) -> None:
        """Wait for the originals' bytes.  A failure of the copy (disk full, a source file gone, permissions) is
        raised HERE, on the main thread, as the reference's copytree would raise it (dataset_balancer.py:70-81):
        a tree of zero-byte placeholders must never be reported as a finished job."""
        t = getattr(self, "_copying", None)
        if t is not None:
            t.join()
            self._copying = None
        err, self._copy_error = getattr(self, "_copy_error", None), None
        if err is not None:
            raise err

    def _images_by_class(self) -> Dict[str, List[Path]]:
        """class name -> its images in the target tree, `*.JPG` before `*.jpg`, each in glob
        order (the order `random.choice` indexes into; dataset_balancer.py:83-93).  Keyed by
        class name only, like the reference."""
        found: Dict[str, List[Path]] = {}
        for plant in self.target_dir.iterdir():
            if not plant.is_dir():
                continue
            for cls in plant.iterdir():
                if cls.is_dir():
                    found[cls.name] = [*cls.glob("*.JPG"), *cls.glob("*.jpg")]
        return found

    def build_tasks(self, images_by_class) -> List[dict]:
        """dataset_balancer.py:105-129 — consumes the global `random` stream in the same order."""
        tasks = []
        src_root, dst_root = str(self.source_dir), str(self.target_dir)
        for class_name, transforms in self.plan.items():
            if class_name not in images_by_class:
                logger.warning(f"No images found for class '{class_name}'")
                continue
            source_images = images_by_class[class_name]
            class_dir_s = str(source_images[0].parent)
            # what a task needs of its source, worked out once per source that is drawn (pathlib's stem / suffix cost
            # more than the draw): (path, path to read from, stem, suffix).  `random.choice(range(n))` draws what
            # `random.choice(source_images)` draws (one `_randbelow(n)`), so the streams are the reference's.
            picks = range(len(source_images))
            info: Dict[int, tuple] = {}
            for transform_name, count in transforms.items():
                for i in range(count):
                    j = random.choice(picks)
                    got = info.get(j)
                    if got is None:
                        source_img = source_images[j]
                        src = str(source_img)   # strings from here on: pathlib costs 10 us per operation, 11,500 times
                        got = info[j] = (src, src_root + src[len(dst_root):] if src.startswith(dst_root + os.sep) else src,
                                         source_img.stem, source_img.suffix)
                    src, read, stem, suffix = got
                    tasks.append({"source_img": src,
                                  "read_img": read,
                                  "output_path": os.path.join(class_dir_s, stem + f"_aug_{transform_name}_{i + 1}" + suffix),
                                  "transform_name": transform_name,
                                  "class_name": class_name,
                                  "seed": random.randint(0, 1000000)})
        return tasks

    # ------------------------------------------------------------------ one share, on one GPU
    def _run_group(self, op: str, images: List[np.ndarray], params: List[dict]) -> List[np.ndarray]:
        """One (transform, size) group as one batched launch on this rank's GPU, host arrays in and
        out (the path for pieces that do not go through the device mirror of the slabs)."""
        import torch
        x = torch.from_numpy(np.stack(images)).cuda()
        res = apply_batch(op, x, params)
        if op != "rotate":  # same-sized results: one device->host copy for the group
            host = torch.stack(res).cpu().numpy()
            return [host[j] for j in range(len(images))]
        return [o.cpu().numpy() for o in res]

    def _gpu_stage(self, chunk: List[dict], decoded: List[tuple], pool: CodecPool, base: int) -> List[tuple]:
        """`decoded[k]` is what the codec worker left for task k: pixels in slot base+k of the input
        slab and the task's parameters (a fresh seeded RNG per task, like the reference's
        `_process_single_transformation`; seed 0 = "unseeded": drawn here from this process's
        global streams).  The chunk's run of the input slab goes to the device as ONE copy (the
        slab is page-locked), the ops run batched by (transform, size) on rows gathered from that
        mirror, results land in a device mirror of the output slab and come back as ONE copy.
        Returns the encode jobs (path, offset, shape, inline array or None)."""
        import torch
        from .. import ops
        n = len(chunk)
        device_path = self._mirror is not None
        t_in = time.perf_counter()
        if device_path:
            dev_in, dev_out = self._mirror
            self._copy_up(pool, base, n, dev_in, decoded)
        groups: Dict[tuple, List[int]] = {}
        images: List[Optional[np.ndarray]] = [None] * n
        params: List[Optional[dict]] = [None] * n
        for k, (task, (status, payload, prm)) in enumerate(zip(chunk, decoded)):
            if status == "err":
                logger.error(f"Failed to process {payload}")
                self.failed += 1
                continue
            if status in ("coef", "scan"):   # pixels do not exist on the host: the GPU finishes the decoding
                img = None
                h, w = payload[0], payload[1]
            else:
                img = pool.view("in", base + k, payload) if status == "ok" else payload
                h, w, _ = img.shape
            if prm is None:
                prm = draw_params(task["transform_name"], w, h)
            elif "noise8" in prm and prm["noise8"] is None:
                # in the (page-locked) noise slab: goes up from there with an asynchronous copy of its own
                prm["noise8"] = pool.tensor("noise", base + k, 1)[0, :h * w * 3].view(h, w, 3) if device_path \
                    else pool.view("noise", base + k, (h, w, 3))
            images[k], params[k] = img, prm
            groups.setdefault((task["transform_name"], h, w, status), []).append(k)
        jobs: List[tuple] = []
        huffman: List[tuple] = []   # (task indexes, device status of the GPU's Huffman decoding) per size
        noise_flags: List[tuple] = []   # (task indexes, device flags of the GPU-made noise planes) per distortion group
        decoded_px: Dict[tuple, tuple] = {}   # (h, w, status) -> (pixels [m, h, w, 3] on the device, task index -> row)
        if device_path:
            # the JPEG back end once per chunk and size, not once per transform: the Huffman kernel's time is a latency
            # (one workgroup per image, ~1 ms whether 40 images or 256), and the IDCT batches better as well
            sizes: Dict[tuple, List[int]] = {}
            for (op, h, w, status), ks in groups.items():
                if status in ("coef", "scan"):
                    sizes.setdefault((h, w, status), []).extend(ks)
            for (h, w, status), ks in sizes.items():
                ks.sort()
                try:
                    rows = dev_in[:n] if len(ks) == n else dev_in[torch.tensor(ks, dtype=torch.int64, device=dev_in.device)]
                    if status == "scan":
                        huffman.append((ks, ops.jpeg_huffman_u8(rows, h, w)))
                    decoded_px[(h, w, status)] = (ops.jpeg_idct_rgb_u8(rows, h, w), {k: j for j, k in enumerate(ks)})
                except Exception as e:  # noqa: BLE001
                    logger.error(f"Failed to decode a batch of {len(ks)} {h}x{w} sources: {e}")
        for (op, h, w, status), ks in groups.items():
            prm = [params[k] for k in ks]
            try:
                if device_path and status in ("ok", "coef", "scan"):
                    idx = torch.tensor(ks, dtype=torch.int64, device=dev_in.device)
                    if status in ("coef", "scan"):
                        px, row = decoded_px[(h, w, status)]
                        x = px[torch.tensor([row[k] for k in ks], dtype=torch.int64, device=dev_in.device)]
                    else:
                        x = dev_in[idx, :h * w * 3].view(len(ks), h, w, 3)
                    if op == "rotate":
                        # the rotated canvases are written by the kernel straight into their slots of the output mirror
                        plan = ops.rotate_expand_plan(w, h, [q["angle"] for q in prm], dev_in.device,
                                                      offsets=[k * pool.slot_bytes for k in ks], limit=pool.slot_bytes)
                        if plan is not None:
                            ops.rotate_expand_apply(x, plan, 255, out=dev_out.view(-1))
                            if all(chunk[k]["output_path"].lower().endswith((".jpg", ".jpeg")) for k in ks):
                                # every canvas has a size of its own: the encoder takes them all in one launch per step
                                # and leaves the finished scans where the pixels were
                                ops.jpeg_encode_items_u8(dev_out.view(-1), [(k * pool.slot_bytes, oh, ow) for k, (oh, ow)
                                                                            in zip(ks, plan["sizes"])], pool.slot_bytes,
                                                         JPEG_QUALITY)
                                jobs += [(chunk[k]["output_path"], (base + k) * pool.slot_bytes, (oh, ow, 3), None, "scan")
                                         for k, (oh, ow) in zip(ks, plan["sizes"])]
                            else:
                                jobs += [(chunk[k]["output_path"], (base + k) * pool.slot_bytes, (oh, ow, 3), None)
                                         for k, (oh, ow) in zip(ks, plan["sizes"])]
                            continue
                    n8 = None
                    if op == "distortion" and all(q is not None and "noise_seed" in q and "noise8" not in q for q in prm):
                        from .image_augmenter import NOISE_LEVEL
                        n8, fl = ops.legacy_normal_u8([q["noise_seed"] for q in prm], 0.0, float(NOISE_LEVEL), h * w * 3,
                                                      dev_in.device)
                        n8 = n8.reshape(len(ks), h, w, 3)
                        noise_flags.append((ks, fl))
                    res = apply_batch(op, x, prm, noise8=n8)
                    if op != "rotate":
                        y = torch.stack(res)
                        # whole-MCU images bound for .jpg files leave the GPU as finished JPEG scans (colour conversion,
                        # downsampling, DCT, quantisation, Huffman coding, byte stuffing): the worker adds the
                        # markers and writes the file
                        jpg = [chunk[k]["output_path"].lower().endswith((".jpg", ".jpeg")) for k in ks]
                        if h % 16 == 0 and w % 16 == 0 and all(jpg):
                            coef = ops.jpeg_fdct_quant_u8(y, JPEG_QUALITY)
                            dev_out[idx] = ops.jpeg_entropy_u8(coef, h, w, out_stride=pool.slot_bytes)
                            kind = "scan"
                        else:
                            dev_out[idx, :h * w * 3] = y.view(len(ks), -1)
                            kind = "px"
                        jobs += [(chunk[k]["output_path"], (base + k) * pool.slot_bytes, (h, w, 3), None, kind)
                                 for k in ks]
                    else:
                        for o, k in zip(res, ks):
                            if o.numel() <= pool.slot_bytes:
                                dev_out[k, :o.numel()] = o.reshape(-1)
                                jobs.append((chunk[k]["output_path"], (base + k) * pool.slot_bytes, tuple(o.shape), None))
                            else:
                                jobs.append((chunk[k]["output_path"], 0, tuple(o.shape), o.cpu().numpy()))
                    continue
                res = self._run_group(op, [images[k] for k in ks], prm)
            except Exception as e:  # noqa: BLE001
                logger.error(f"Failed batch {op}: {e}")
                self.failed += len(ks)
                continue
            for r, k in zip(res, ks):
                if r.nbytes <= pool.slot_bytes:
                    if device_path:   # joins the chunk's single copy back
                        dev_out[k, :r.nbytes] = torch.from_numpy(np.ascontiguousarray(r)).reshape(-1).cuda()
                    else:
                        pool.view("out", base + k, r.shape)[...] = r
                    jobs.append((chunk[k]["output_path"], (base + k) * pool.slot_bytes, tuple(r.shape), None))
                else:
                    jobs.append((chunk[k]["output_path"], 0, tuple(r.shape), r))
        t_ops = time.perf_counter()
        if device_path:
            self._copy_back(pool, base, n, dev_out, jobs)   # waits for it: the encoders may start
        redo: List[int] = []
        for ks, st in huffman + noise_flags:
            st = st.cpu().numpy()
            if st.any():   # scans the GPU could not decode (damaged files, mostly): libjpeg has the reference's verdict;
                redo += [k for k, v in zip(ks, st) if v and k not in redo]   # noise planes a last bit of log() could change
        if redo:
            jobs = self._redo_on_host(chunk, redo, params, pool, base, jobs)
        t_out = time.perf_counter()
        self.timings["gpu_stage_host_ops"] = self.timings.get("gpu_stage_host_ops", 0.0) + (t_ops - t_in)
        self.timings["gpu_stage_sync_d2h"] = self.timings.get("gpu_stage_sync_d2h", 0.0) + (t_out - t_ops)
        return jobs

    def _copy_up(self, pool: CodecPool, base: int, n: int, dev_in, decoded: List[tuple]) -> None:
        """The chunk's run of the INPUT slab -> its device mirror.  When every source arrived as a prepared scan (the
        usual case) only the parts of the slots that hold something cross PCIe — the quantisation tables at the front,
        and header / Huffman tables / scan behind the (device-only) coefficient area: ~27 KB of a 301 KB slot for a
        224 x 224 file — as two row copies (hipMemcpy2DAsync between the page-locked slab and the mirror; torch's own
        strided copy goes through a pageable temporary and measured 8x slower than copying whole slots).  Otherwise:
        whole slots, one contiguous copy."""
        import torch
        from .. import _lib
        from ..utils import jpeg_host
        host = pool.tensor("in", base, n)
        live = [d for d in decoded if d[0] != "err"]
        if live and all(d[0] == "scan" for d in live):
            lo = min(jpeg_host.scan_aux_offset(h, w) for h, w in {(d[1][0], d[1][1]) for d in live})
            hi = (max(d[1][3] for d in live) + 15) // 16 * 16
            if 256 <= lo < hi <= pool.slot_bytes:
                stream = torch.cuda.current_stream().cuda_stream
                slot = pool.slot_bytes
                _lib.call("lf_copy_rows", dev_in.data_ptr(), slot, host.data_ptr(), slot, 256, n, 0, stream)
                _lib.call("lf_copy_rows", dev_in.data_ptr() + lo, slot, host.data_ptr() + lo, slot, hi - lo, n, 0, stream)
                return
        dev_in[:n].copy_(host, non_blocking=True)

    def _copy_back(self, pool: CodecPool, base: int, n: int, dev_out, jobs: List[tuple]) -> None:
        """The device mirror of the OUTPUT slab -> the chunk's run of the slab, and wait for it.  The front of every slot
        in one row copy (wide enough for the finished JPEG scans seen so far, +25 %), the rest of the slots whose
        content is known to be longer (rotated canvases, pixel results) row by row; a scan that turns out longer than
        the guess is fetched in a second step."""
        import torch
        from .. import _lib
        slot = pool.slot_bytes
        host = pool.tensor("out", base, n)
        cur = torch.cuda.current_stream()
        width = min(slot, getattr(self, "_scan_width", 0) or max(4096, (slot // 8 + 4095) // 4096 * 4096))
        _lib.call("lf_copy_rows", host.data_ptr(), slot, dev_out.data_ptr(), slot, width, n, 1, cur.cuda_stream)
        scans: List[int] = []
        for j in jobs:
            if j[3] is not None:
                continue
            k = j[1] // slot - base
            if len(j) > 4 and j[4] == "scan":
                scans.append(k)
                continue
            need = int(np.prod(j[2]))
            if need > width:
                host[k, width:need].copy_(dev_out[k, width:need], non_blocking=True)
        cur.synchronize()
        if scans:
            lens = np.frombuffer(pool.slabs["out"].buf, np.int32, n * (slot // 4), base * slot)[::slot // 4]
            longest = 0
            more = False
            for k in scans:
                nbytes = int(lens[k]) + 4
                longest = max(longest, nbytes)
                if nbytes > width:
                    host[k, width:nbytes].copy_(dev_out[k, width:nbytes], non_blocking=True)
                    more = True
            if more:
                cur.synchronize()
            self._scan_width = max(getattr(self, "_scan_width", 0), min(slot, (longest * 5 // 4 + 4095) // 4096 * 4096))

    def _redo_on_host(self, chunk: List[dict], ks: List[int], params: List[Optional[dict]], pool: CodecPool, base: int,
                      jobs: List[tuple]) -> List[tuple]:
        """Tasks whose source the GPU's Huffman decoder handed back: the file goes through Pillow here, as it would have
        in the reference's worker (image_utils.py:19-33) — pixels or an error — and the op runs on what Pillow gives.
        Their slots of the OUTPUT slab are written after the chunk's copy back; their encode jobs are replaced."""
        from ..utils.image_utils import ImageLoader
        paths = {chunk[k]["output_path"] for k in ks}
        jobs = [j for j in jobs if j[0] not in paths]
        for k in ks:
            task = chunk[k]
            try:
                img = ImageLoader.load_as_array(task.get("read_img", task["source_img"]))
                prm = params[k]
                if prm is not None and "noise8" in prm and prm["noise8"] is not None and tuple(prm["noise8"].shape) != img.shape:
                    raise ValueError("the file's pixels are not of the size its header gave")
                r = self._run_group(task["transform_name"], [img], [prm])[0]
            except Exception as e:  # noqa: BLE001 — the reference counts any failure
                logger.error(f"Failed to process {task['source_img']} - {e}")
                self.failed += 1
                continue
            if r.nbytes <= pool.slot_bytes:
                pool.view("out", base + k, r.shape)[...] = r
                jobs.append((task["output_path"], (base + k) * pool.slot_bytes, tuple(r.shape), None))
            else:
                jobs.append((task["output_path"], 0, tuple(r.shape), r))
        return jobs

    def _collect(self, futures, paths: List[str]) -> None:
        flags = [ok for f in futures for ok in f.result()]
        for ok, path in zip(flags, paths):
            if ok:
                self.completed += 1
            else:
                self.failed += 1
                logger.error(f"Failed: {path}")

    def _run_share(self, share: List[dict], total: int) -> None:
        """decode | kernels | encode over chunks of this rank's share: chunk i+1 is being decoded and
        chunk i-1 encoded (codec worker processes, pixels through shared memory) while chunk i is on
        the GPU.  Slots are sized from the first source image (x2: a 30-degree rotation grows the
        canvas 1.87x); anything larger travels as a pickled array."""
        if not share:
            if self._codec is not None:
                self._codec.close()
                self._codec = None
            return
        from PIL import Image
        n_chunk = min(CHUNK, len(share))
        chunks = [share[b:b + n_chunk] for b in range(0, len(share), n_chunk)]
        with Image.open(share[0].get("read_img", share[0]["source_img"])) as probe:
            w0, h0 = probe.size
        slot = (2 * h0 * w0 * 3 + 4095) // 4096 * 4096
        t0 = time.perf_counter()
        pool = self._codec or CodecPool(self.workers)
        pool.allocate(RING * n_chunk, slot)
        self._mirror = None
        try:
            import torch
            if type(self)._run_group is DatasetBalancer._run_group and torch.cuda.is_available() and pool.pin():
                dev = torch.device("cuda", torch.cuda.current_device())
                self._mirror = (torch.empty((n_chunk, slot), dtype=torch.uint8, device=dev),
                                torch.empty((n_chunk, slot), dtype=torch.uint8, device=dev))
            # 0: the workers decode to pixels; 1: they Huffman-decode only, IDCT .. colour on the GPU; 2: they read the
            # markers only, the Huffman decoding is the GPU's too (LEAFFLICTION_GPU_HUFFMAN=0: stay with 1)
            gpu_decode = 0 if self._mirror is None else (1 if os.environ.get("LEAFFLICTION_GPU_HUFFMAN", "1") == "0" else 2)
            # chunk i + 2 is queued for decoding before chunk i goes to the GPU: the workers always have a
            # chunk's worth of work behind the one the main thread is waiting for
            # the distortion tasks' noise planes (RandomState(seed).normal: 2.6 ms of a worker's time each) on the GPU too
            gpu_noise = self._mirror is not None and os.environ.get("LEAFFLICTION_GPU_NOISE", "1") != "0"
            # pieces per worker: four while a piece may hold several distortion tasks whose noise the worker makes, one
            # when all the workers do per task is read a file and its markers (every future costs the parent ~20 us)
            pieces = 1 if gpu_noise and gpu_decode == 2 else 4
            ahead = [pool.decode(chunks[0], 0, gpu_decode, pieces, gpu_noise=gpu_noise)]
            ahead[0][0].result()   # the workers are up (spawn + imports) once the first piece is back
            self.timings["codec_pool_start"] = time.perf_counter() - t0
            if len(chunks) > 1:
                ahead.append(pool.decode(chunks[1], n_chunk, gpu_decode, pieces, gpu_noise=gpu_noise))
            encoding, enc_paths = [], []
            decoded = jobs = None
            tw = {"wait_decode": 0.0, "gpu_stage": 0.0, "wait_encode": 0.0}
            self.timings.update(tw)
            self.timings["slabs_page_locked"] = float(self._mirror is not None)
            for i, chunk in enumerate(chunks):
                ta = time.perf_counter()
                decoded = [r for f in ahead.pop(0) for r in f.result()]
                if i + 2 < len(chunks):
                    ahead.append(pool.decode(chunks[i + 2], ((i + 2) % RING) * n_chunk, gpu_decode, pieces, gpu_noise=gpu_noise))
                tb = time.perf_counter()
                jobs = self._gpu_stage(chunk, decoded, pool, (i % RING) * n_chunk)
                tc = time.perf_counter()
                self._collect(encoding, enc_paths)
                td = time.perf_counter()
                self.timings["wait_decode"] += tb - ta
                self.timings["gpu_stage"] += tc - tb
                self.timings["wait_encode"] += td - tc
                encoding, enc_paths = pool.encode(jobs), [j[0] for j in jobs]
                done = self.completed + self.failed
                if done and done % 500 < n_chunk:
                    logger.info(f"Progress (rank {self.ranks.rank}): {done}/{len(share)} of this share "
                                f"({total} tasks in all) - {self.completed} success, {self.failed} failed")
            self._collect(encoding, enc_paths)
            # what is left of the originals' copy now has every worker (the thread is joined, and its failure raised,
            # by _join_copy)
            tj = time.perf_counter()
            self._copy_boost = True
            t = getattr(self, "_copying", None)
            if t is not None:
                t.join()
            self.timings["copy_originals_tail"] = time.perf_counter() - tj
        finally:
            decoded = jobs = None   # the last views of the slabs
            self._mirror = None
            self._codec = None
            pool.close()

    # ------------------------------------------------------------------ the whole job
    def execute_balancing(self):
        if not self.plan:
            logger.info("No augmentation plan - skipping execution")
            return
        rk = self.ranks
        t0 = time.perf_counter()
        self._codec = CodecPool(self.workers)   # the workers start (and import) while the tree is copied
        try:
            self._execute_shares(rk, t0)
        except BaseException:
            # the path that never reached _run_share's own clean-up (or left the copy thread behind): stop the codec
            # workers and take their shared-memory slabs down now rather than leave that to exit handlers, and wait
            # for the copy thread — its own failure, if any, yields to the exception already on its way up
            pool, self._codec = getattr(self, "_codec", None), None
            if pool is not None:
                pool.close()
            try:
                self._join_copy()
            except BaseException:  # noqa: BLE001
                pass
            raise

    def _execute_shares(self, rk, t0: float) -> None:
        prep_error = None
        if rk.rank == 0:
            # A failure while rank 0 prepares the job (source missing, rmtree / permission errors, a bad plan) must
            # reach EVERY rank: the others would otherwise wait in the broadcast until the process group times out.
            try:
                self._fresh_target()
                t1 = time.perf_counter()
                self.tasks = self.build_tasks(self._images_by_class())
                self.timings = {"copy_originals": t1 - t0, "task_list": time.perf_counter() - t1}
            except Exception as e:  # noqa: BLE001
                prep_error = e
        payload = rk.broadcast_object(({"error": f"{type(prep_error).__name__}: {prep_error}"} if prep_error is not None
                                       else {"tasks": self.tasks}) if rk.rank == 0 else None)
        if "error" in payload:
            if prep_error is not None:
                raise prep_error
            raise RuntimeError(f"rank 0 could not prepare the balancing job - {payload['error']}")
        self.tasks = payload["tasks"]
        total = len(self.tasks)
        begin, end = contiguous_share(total, rk.rank, rk.world)
        logger.info(f"Starting GPU augmentation: {total} images to generate"
                    + (f" (rank {rk.rank}/{rk.world}: tasks {begin}..{end - 1})" if rk.active else ""))
        streams = (random.getstate(), np.random.get_state())
        t2 = time.perf_counter()
        self._run_share(self.tasks[begin:end], total)
        self.timings["decode_kernels_encode"] = time.perf_counter() - t2 - self.timings.get("copy_originals_tail", 0.0)
        random.setstate(streams[0])
        np.random.set_state(streams[1])
        self.completed, self.failed = rk.sum_ints([self.completed, self.failed])
        rk.barrier()  # every share's files are on disk before the tree is listed
        logger.info(f"Augmentation complete: {self.completed} images generated, {self.failed} failed")
        t3 = time.perf_counter()
        self._join_copy()
        self.timings["copy_originals_tail"] = self.timings.get("copy_originals_tail", 0.0) + time.perf_counter() - t3
        rk.barrier()
        t3 = time.perf_counter()
        if rk.rank == 0:
            self._generate_augmented_manifest()
        rk.barrier()
        self.timings["manifest"] = time.perf_counter() - t3

    def _generate_augmented_manifest(self):
        self.manifest_generator = ManifestGenerator(self.analyzer.original_manifest, self.source_dir,
                                                    self.target_dir, self.workers)
        out_dir = self.manifest_path.parent if self.manifest_path is not None else Path("artifacts/datasets")
        out_dir.mkdir(parents=True, exist_ok=True)
        # = save_manifest(generate_augmented_manifest(), ...), byte for byte, without 172,000 dictionaries on the way
        self.manifest_generator.write_augmented_manifest(out_dir / "manifest_augmented.json")

    def run(self):
        logger.info("=== Dataset Balancing System ===")
        try:
            self.analyze_distribution()
            self.calculate_plan()
            self.execute_balancing()
            logger.info("=== Balancing Complete ===")
        except Exception as e:
            logger.error(f"Dataset balancing failed - {e}")
            raise
