#!/usr/bin/env python3
"""Headline benchmark: leaf_cnn training images/sec on MI355X (BASELINE.json config C2).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full training step of the reference's `srcs/cli/train.py` path on one
per-GPU batch of synthetic input that is already resident in HBM: fused input stage
(uint8 HWC -> augment -> Normalization -> f32 NCHW), leaf_cnn `base` forward, label-smoothed
cross-entropy, backward, (N>1: one RCCL all-reduce of the flat gradient bucket), AdamW with
per-tensor clipnorm + cosine LR, EMA.  fp32 throughout (BASELINE.json configs[1]); weak
scaling: every rank keeps batch 256.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel,
algorithmic FLOP / measured launch duration from HIP events recorded inside the timed
region) and `cpu_baseline` (the torch-CPU oracle's training step on the host cores).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from collections import defaultdict
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WIDTHS = [32, 64, 128, 256]
NUM_CLASSES = 8
IMG = 224
BATCH = 256
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
TRAIN_GFLOP_PER_IMG = 18.757  # 3 x 3.126 GMAC x 2

FWD_NAMES = {0: "conv_mfma_kernel<T,32,8,1,1,4,2>", 1: "conv_mfma_kernel<T,32,8,1,2,4,2>",
             2: "conv_mfma_kernel<T,16,16,1,1,4,2>", 3: "conv_mfma_kernel<T,16,16,1,2,4,2>",
             4: "conv_mfma_kernel<T,28,8,4,1,1,7>", 5: "conv_mfma_kernel<T,32,8,2,2,2,4>",
             6: "conv_mfma_kernel<T,56,8,2,1,2,7>"}
WG_NAMES3 = {0: "wgrad3_kernel<32,4,1,1,4>", 1: "wgrad3_kernel<16,8,1,2,2>",
             2: "wgrad3_kernel<16,4,2,2,1>", 3: "wgrad3_kernel<28,2,2,2,1>",
             4: "wgrad3_kernel<32,4,1,2,2>", 5: "wgrad_smallcin_kernel<32,8>"}
WG_NAMES1 = {0: "wgrad_mfma_kernel<1,32,4,1,1,4>", 1: "wgrad_mfma_kernel<1,16,8,1,2,2>",
             2: "wgrad_mfma_kernel<1,16,4,2,2,1>", 3: "wgrad_mfma_kernel<1,28,2,2,2,1>",
             4: "wgrad_mfma_kernel<1,32,4,1,2,2>"}


class KernelTimer:
    """HIP events around the conv launches, on the stream they are launched on."""

    def __init__(self, lib_mod):
        self.lib_mod = lib_mod
        self.records = []
        self.enabled = False
        self._orig = lib_mod.call

    def install(self):
        timer = self

        def call(name, *args):
            if not timer.enabled or name not in ("lf_conv2d_f32", "lf_conv2d_stats_f32", "lf_conv2d_bnbwd_f32",
                                                 "lf_conv2d_wgrad_f32", "lf_conv2d_wgrad_bn_f32",
                                                 "lf_conv2d_bf16_train", "lf_conv2d_wgrad_bf16"):
                return timer._orig(name, *args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = timer._orig(name, *args)
            e1.record()
            lib = timer.lib_mod.load()
            if name == "lf_conv2d_bf16_train":
                n, cin, h, w, cout, k = args[4:10]
                px = n * h * w
                stem = (not args[1]) and cin <= 3 and k == 3 and cout == 32
                if cout in (32, 64) and w % 8 == 0 and (stem or (args[1] and cin in (32, 64))) and \
                        (k == 3 or (cin, cout) in ((32, 64), (64, 32))):
                    tw, th = (64, 4) if w >= 64 else ((32, 8) if w % 32 == 0 else (16, 16))   # plan_s of lf_conv_bf16s.hip
                    kname = f"conv_bf16s_kernel<{k * k},{16 if stem else cin},{cout // 32},{tw},{th}> {cin}->{cout}@{h}"
                else:
                    kname = f"conv_bf16_kernel<{k * k},{'2,2' if cout % 64 == 0 else '1,4'}> {cin}->{cout}@{h}"
                nbytes = px * ((2.0 if args[1] else 4.0) * cin + 2.0 * cout) + 2.0 * cin * cout * k * k
                nbytes += 2.0 * px * cout * ((1 if args[13] else 0) + (1 if args[17] else 0))
                timer.records.append((kname, 2.0 * px * cin * cout * k * k, e0, e1, nbytes))
                return rc
            if name == "lf_conv2d_wgrad_bf16":
                n, cin, h, w, cout, k = args[9:15]
                px = n * h * w
                kname = f"wgrad_bf16_kernel<{k * k}> {cin}->{cout}@{h}"
                nbytes = px * ((4.0 if cin * 9 <= 32 and k == 3 else 2.0) * cin + 2.0 * cout) + 4.0 * cin * cout * k * k
                nbytes += 2.0 * px * cout * ((1 if args[2] else 0) + (1 if args[7] else 0))
                timer.records.append((kname, 2.0 * px * cin * cout * k * k, e0, e1, nbytes))
                return rc
            if name in ("lf_conv2d_f32", "lf_conv2d_stats_f32", "lf_conv2d_bnbwd_f32"):
                n, cin, h, w, cout, k = args[3:9]
                kname = FWD_NAMES[lib.lf_conv2d_variant(h, w, cout, k)].replace("T", str(k * k))
            else:
                n, cin, h, w, cout, k = args[8:14] if name == "lf_conv2d_wgrad_bn_f32" else args[2:8]
                kname = (WG_NAMES3 if k == 3 else WG_NAMES1)[lib.lf_conv2d_wgrad_variant(n, cin, h, w, cout, k)]
            flop = 2.0 * n * h * w * cin * cout * k * k
            # operands once: both activation tensors + the weights / weight-gradient
            nbytes = 4.0 * (n * h * w * (cin + cout) + cin * cout * k * k)
            if name == "lf_conv2d_wgrad_bn_f32":  # also reads the BN input and writes dy
                nbytes += 8.0 * n * h * w * cout
            elif name == "lf_conv2d_bnbwd_f32":     # also reads the BN input for the mask ...
                nbytes += 4.0 * n * h * w * cout * (2 if args[9] else 1)   # ... and y when accumulating
            elif name == "lf_conv2d_f32" and args[12]:  # accumulate: y is read and written
                nbytes += 4.0 * n * h * w * cout
            timer.records.append((kname, flop, e0, e1, nbytes))
            return rc

        self.lib_mod.call = call
        # nn.py / ops.py bound `_lib` as a module, so patching the attribute is enough

    def summary(self):
        torch.cuda.synchronize()
        agg = defaultdict(lambda: [0.0, 0.0, 0, 0.0])
        for kname, flop, e0, e1, nbytes in self.records:
            a = agg[kname]
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += flop
            a[2] += 1
            a[3] += nbytes
        return {k: {"seconds": v[0], "flop": v[1], "launches": v[2], "bytes": v[3]}
                for k, v in agg.items()}


def usable_cores() -> int:
    """Cores this process may actually use (affinity mask capped by the cgroup CPU quota)."""
    from leaffliction_amd.utils.system_info import get_available_cores
    return get_available_cores()


def cpu_baseline(seconds_budget: float = 25.0):
    """The oracle (torch CPU fp32 restatement of the reference's Keras train step) timed on
    this box's host cores: base preset, img 224, batch 32 (the reference default, train.py:67)."""
    from oracle import cnn_ref as R
    threads = usable_cores()
    torch.set_num_threads(threads)   # not torch's default (every core it can see): no oversubscription
    bs = 32
    params = R.init_params(NUM_CLASSES, WIDTHS, seed=0)
    state = R.init_state(WIDTHS)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((bs, 3, IMG, IMG), generator=g)
    y = torch.nn.functional.one_hot(torch.randint(0, NUM_CLASSES, (bs,), generator=g), NUM_CLASSES).float()
    drops = [torch.ones(bs, f) for f in WIDTHS]
    top = torch.ones(bs, WIDTHS[-1])
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in params.items()}
    steps, t_used = 0, 0.0
    R.train_step(params, state, x, y, WIDTHS, drops, top)  # warm-up (allocations, threads)
    while steps < 1 or (t_used < seconds_budget and steps < 8):
        t0 = time.perf_counter()
        _l, _d, _p, grads = R.train_step(params, state, x, y, WIDTHS, drops, top)
        params, m, v = R.adamw_step(params, grads, m, v, steps + 1, 2e-3)
        t_used += time.perf_counter() - t0
        steps += 1
    return {"value": round(bs * steps / t_used, 2), "unit": "images/sec", "cores": threads,
            "gflops": round(TRAIN_GFLOP_PER_IMG * bs * steps / t_used, 1),
            "kind": "port",
            "sample": f"{steps} training steps of leaf_cnn base at 224x224, batch {bs}, fp32 "
                      f"(oracle/cnn_ref.py on torch CPU, {threads} threads)"}


def cpu_augment_baseline(images: int = 24):
    """The same six operations done by the libraries the reference calls (Pillow + numpy, the
    calls of image_augmenter.py:20-133 on in-memory 224x224 images, no JPEG codec), one host core:
    the CPU figure beside the augmentation-pass table.  A reported baseline, not a target."""
    import numpy as np
    from PIL import Image, ImageOps
    rng = np.random.RandomState(7)
    imgs = [Image.fromarray(rng.randint(0, 256, (IMG, IMG, 3)).astype(np.uint8)) for _ in range(images)]
    W = H = IMG

    def op_flip(im):
        return im.transpose(Image.FLIP_LEFT_RIGHT if rng.rand() < 0.5 else Image.FLIP_TOP_BOTTOM)

    def op_rotate(im):
        return im.rotate(rng.uniform(-30, 30), expand=True, fillcolor="white")

    def op_skew(im):
        f = rng.uniform(0.05, 0.15)
        return im.transform((W, H), Image.PERSPECTIVE, [1 + f, 0, -f * W, 0, 1 + f, -f * H, 0, 0], Image.BICUBIC)

    def op_shear(im):
        v = rng.uniform(-0.2, 0.2)
        return im.transform((W, H), Image.AFFINE, [1, v, 0, 0, 1, 0] if rng.rand() < 0.5 else [1, 0, 0, v, 1, 0],
                            Image.BICUBIC)

    def op_crop(im):
        r = rng.uniform(0.8, 0.95)
        nw, nh = int(W * r), int(H * r)
        left, top = rng.randint(0, W - nw + 1), rng.randint(0, H - nh + 1)
        return im.crop((left, top, left + nw, top + nh)).resize((W, H), Image.LANCZOS)

    def op_distortion(im):
        a = np.array(im)
        noisy = a + rng.normal(0, 5, a.shape).astype(np.uint8)
        return ImageOps.autocontrast(Image.fromarray(noisy), cutoff=rng.uniform(0, 2))

    per_op, inv = {}, 0.0
    for name, fn in (("flip", op_flip), ("rotate", op_rotate), ("skew", op_skew), ("shear", op_shear),
                     ("crop", op_crop), ("distortion", op_distortion)):
        fn(imgs[0])
        t0 = time.perf_counter()
        for im in imgs:
            fn(im)
        sec = (time.perf_counter() - t0) / images
        per_op[name] = round(1.0 / sec, 1)
        inv += sec / 6.0
    return {"mix_images_per_sec": round(1.0 / inv, 1), "per_op_images_per_sec": per_op, "cores": 1,
            "kind": "the reference's libraries (Pillow/numpy calls of image_augmenter.py), in memory",
            "sample": f"{images} synthetic 224x224 images per operation"}


def augment_throughput(dev, n=4096, iters=5, only=None):
    """Second half of the headline metric: the augmentation pass on synthetic 224x224x3 images
    resident in HBM (BASELINE configs[2]; op mix 1/6 each like the balancer's plan).  Host-side
    parameter tables are built outside the timed region (they are inputs); JPEG decode/encode
    is excluded.  GB/s = algorithmic bytes (input once + output once) / time."""
    import numpy as np

    from leaffliction_amd import ops
    g = torch.Generator().manual_seed(42)
    x = torch.randint(0, 256, (n, IMG, IMG, 3), dtype=torch.uint8, generator=g).to(dev)
    rng = np.random.RandomState(42)
    img_b = IMG * IMG * 3
    mode = torch.from_numpy(rng.randint(0, 2, n).astype(np.int32)).to(dev)
    f = rng.uniform(0.05, 0.15, n)
    skew = torch.tensor([[1 + v, 0, -v * IMG, 0, 1 + v, -v * IMG, 0, 0] for v in f],
                        dtype=torch.float64, device=dev)
    sh = rng.uniform(-0.2, 0.2, n)
    shear = torch.tensor([[1, v, 0, 0, 1, 0, 0, 0] if i % 2 else [1, 0, 0, v, 1, 0, 0, 0]
                          for i, v in enumerate(sh)], dtype=torch.float64, device=dev)
    angles = rng.uniform(-30, 30, n)
    rplan = ops.rotate_expand_plan(IMG, IMG, angles, dev)
    rbuf = torch.empty(rplan["total"], dtype=torch.uint8, device=dev)
    boxes = []
    for _ in range(n):
        r = rng.uniform(0.8, 0.95)
        nw = nh = int(IMG * r)
        boxes.append((rng.randint(0, IMG - nw + 1), rng.randint(0, IMG - nh + 1), nw, nh))
    ctab = ops.crop_resize_plan(IMG, IMG, boxes, dev)
    cut = torch.from_numpy(rng.uniform(0, 2, n)).to(dev)
    rot_out_b = sum(a * b * 3 for a, b in rplan["sizes"]) / n
    mask = (torch.rand((n, IMG, IMG), generator=g) > 0.4).to(torch.uint8).mul_(255).to(dev)
    cases = {
        "flip": (lambda: ops.flip_u8(x, mode), 2 * img_b),
        "rotate": (lambda: ops.rotate_expand_apply(x, rplan, 255, rbuf), img_b + rot_out_b),
        "skew": (lambda: ops.warp_bicubic_u8(x, skew, True, True), 2 * img_b),
        "shear": (lambda: ops.warp_bicubic_u8(x, shear, False), 2 * img_b),
        "crop": (lambda: ops.resample_u8(x, IMG, IMG, ctab[0], ctab[1], ctab[2], ctab[3], True, ctab[4]), 2 * img_b),
        # distortion = noise add (2 passes) + histogram (1 read) + LUT apply (2): 5 image passes by the reference's
        # count (kept as the algorithmic figure); the kernels make 4 — the histogram is taken while the noise is added
        "distortion": (lambda: ops.distortion_u8(x, cut, seed=42, sigma=5.0), 5 * img_b),
        "pack": (lambda: ops.pack_hwc_u8_to_nchw_f32(x), img_b + 4 * img_b),
        "hist": (lambda: ops.hist_u8(x), img_b + 3072),
        # the transform-side kernels the north star names: mask-and-composite, separable blur,
        # colour-space conversion (image + mask in, image out; gray blur = 1 plane each way)
        "mask_composite": (lambda: ops.mask_composite_u8(x, mask), 2 * img_b + IMG * IMG),
        "gauss_blur_5x5": (lambda: ops.gauss_blur_u8(x, 5, 1.5), 2 * img_b),
        "gauss_blur_15x15": (lambda: ops.gauss_blur_u8(x, 15, 0.0), 2 * img_b),
        "rgb2hsv": (lambda: ops.rgb2hsv_u8(x), 2 * img_b),
        # the whole saliency filter of transform/filters/blur.py (image + leaf mask in, image out)
        "blur_saliency": (lambda: ops.blur_saliency_u8(x, mask), 2 * img_b + IMG * IMG),
        # make_mask's default candidate mask (mask.py:727-831): image in, one byte plane out
        "inclusive_mask": (lambda: ops.inclusive_mask_u8(x), img_b + IMG * IMG),
    }
    out, inv = {}, 0.0
    if only:
        cases = {k: v for k, v in cases.items() if k in only}
    for name, (fn, nbytes) in cases.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / iters
        out[name] = {"images_per_sec": round(n / sec), "GB_s": round(n * nbytes / sec / 1e9, 1),
                     "frac_hbm_8TBs": round(n * nbytes / sec / 8e12, 3)}
        if name in ("flip", "rotate", "skew", "shear", "crop", "distortion"):
            inv += sec / n / 6.0
    out["mix_images_per_sec"] = round(1.0 / inv) if inv else None
    out["n_images"] = n
    return out


def codec_edge_throughput(dev, n=256):
    """The kernels at the JPEG edge of the balancer, at one chunk's size (256 tasks): Huffman decoding of the sources
    (Pillow-written 224x224 quality-95 files; the host reader it replaces timed beside it on one core), the noise planes
    of the chunk's distortion tasks (numpy's legacy stream), and the encoder over rotated canvases of different sizes."""
    import io

    import numpy as np
    from PIL import Image

    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    rng = np.random.RandomState(7)
    yy, xx = np.mgrid[0:IMG, 0:IMG]
    files = []
    for i in range(n):
        img = np.stack([128 + 100 * np.sin(xx / 17.0 + i) * np.cos(yy / 23.0), 90 + 80 * np.cos(xx / 9.0),
                        140 + 60 * np.sin((xx + yy) / 31.0)], -1) + rng.normal(0, 3 + 4 * (i % 3), (IMG, IMG, 3))
        b = io.BytesIO()
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(b, format="JPEG", quality=95)
        files.append(b.getvalue())
    stride = (2 * IMG * IMG * 3 + 4095) // 4096 * 4096
    slots = np.zeros((n, stride), np.uint8)
    for i, f in enumerate(files):
        if jpeg_host.scan_prepare_into(f, slots[i]) is None:
            raise RuntimeError("codec_edge: a Pillow-written file was not taken")
    ref = np.zeros(stride, np.uint8)
    t0 = time.perf_counter()
    for f in files[:64]:
        jpeg_host.read_file_into(f, ref)
    host = 64 / (time.perf_counter() - t0)
    d = torch.from_numpy(slots).to(dev)

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps
    sec = timed(lambda: ops.jpeg_huffman_u8(d, IMG, IMG))
    if int(ops.jpeg_huffman_u8(d, IMG, IMG).abs().sum()):
        raise RuntimeError("codec_edge: the GPU decoder handed a file back")
    out = {"huffman_decode": {"files_per_sec": round(n / sec), "ms_per_launch": round(sec * 1e3, 3), "files": n,
                              "mean_file_bytes": round(float(np.mean([len(f) for f in files]))),
                              "host_reader_files_per_sec_one_core": round(host)}}
    seeds = list(range(1000, 1000 + n // 6))
    sec = timed(lambda: ops.legacy_normal_u8(seeds, 0.0, 5.0, IMG * IMG * 3, dev))
    out["noise_planes"] = {"planes_per_sec": round(len(seeds) / sec), "ms_per_launch": round(sec * 1e3, 3),
                           "planes": len(seeds), "host_ms_per_plane_one_core": 2.6}
    sizes = [(int(a), int(b)) for a, b in zip(rng.randint(IMG, 307, n // 6), rng.randint(IMG, 307, n // 6))]
    canv = torch.randint(0, 256, (len(sizes), stride), dtype=torch.uint8, device=dev)
    items = [(i * stride, a, b) for i, (a, b) in enumerate(sizes)]
    work = torch.empty_like(canv)

    def enc():
        work.copy_(canv)
        ops.jpeg_encode_items_u8(work.view(-1), items, stride, 95)
    sec = timed(enc) - timed(lambda: work.copy_(canv))
    out["encode_mixed_sizes"] = {"images_per_sec": round(len(sizes) / sec), "ms_per_launch_pair": round(sec * 1e3, 3),
                                 "images": len(sizes), "content": "uniform noise (the longest scans)"}
    return out


# ---- end to end: `Augmentation.py` over synthetic 224x224 JPEGs (BASELINE configs[2]) -----------
# class counts per 1,000 generated images: the balancer tops every class of a plant up to the plant's largest
# (x 100 = BASELINE configs[2]'s 100,000 generated files from 72,000 originals)
_E2E_UNIT = {"Apple": {"healthy": 215, "rust": 45, "scab": 45, "rot": 45},
             "Grape": {"healthy": 215, "esca": 85, "spot": 35, "blight": 35}}


def _e2e_layout(generated: int):
    k = max(1, round(generated / 1000))
    return {plant: {cls: n * k for cls, n in classes.items()} for plant, classes in _E2E_UNIT.items()}


def _e2e_write(job):
    from PIL import Image
    arr, path = job
    Image.fromarray(arr).save(path, quality=95)
    return True


def _e2e_make_dataset(root: Path, dev, threads: int, layout) -> int:
    """The synthetic originals: low-frequency colour fields + noise (JPEG sizes like photographs:
    ~25 KB), generated on the GPU in chunks, encoded on host threads.  Not timed."""
    from concurrent.futures import ThreadPoolExecutor
    g = torch.Generator(device=dev).manual_seed(42)
    jobs = []
    for plant, classes in layout.items():
        for cls, n in classes.items():
            d = root / plant / f"{plant}_{cls}"
            d.mkdir(parents=True)
            jobs += [d / f"image ({i + 1}).JPG" for i in range(n)]
    with ThreadPoolExecutor(max_workers=threads) as pool:
        for b0 in range(0, len(jobs), 512):
            paths = jobs[b0:b0 + 512]
            low = torch.rand((len(paths), 3, 14, 14), generator=g, device=dev)
            img = torch.nn.functional.interpolate(low, size=(IMG, IMG), mode="bilinear", align_corners=False)
            img = (img * 255 + torch.randn(img.shape, generator=g, device=dev) * 8).clamp_(0, 255)
            arr = img.permute(0, 2, 3, 1).to(torch.uint8).cpu().numpy()
            list(pool.map(_e2e_write, [(arr[i], paths[i]) for i in range(len(paths))]))
    return len(jobs)


def _pil_task(task):
    """One task the way the reference's pool worker does it (image_augmenter.py:20-133): Pillow decode ->
    one of the six operations with the task's seed -> Pillow encode, quality 95."""
    import random

    import numpy as np
    from PIL import Image, ImageOps
    src, dst, op, seed = task
    random.seed(seed)
    np.random.seed(seed)
    im = Image.open(src).convert("RGB")
    W, H = im.size
    if op == "flip":
        out = im.transpose(Image.FLIP_LEFT_RIGHT if random.choice([True, False]) else Image.FLIP_TOP_BOTTOM)
    elif op == "rotate":
        out = im.rotate(random.uniform(-30, 30), expand=True, fillcolor="white")
    elif op == "skew":
        f = random.uniform(0.05, 0.15)
        out = im.transform((W, H), Image.PERSPECTIVE, [1 + f, 0, -f * W, 0, 1 + f, -f * H, 0, 0], Image.BICUBIC)
    elif op == "shear":
        v = random.uniform(-0.2, 0.2)
        out = im.transform((W, H), Image.AFFINE, [1, v, 0, 0, 1, 0] if random.choice([True, False]) else [1, 0, 0, v, 1, 0],
                           Image.BICUBIC)
    elif op == "crop":
        r = random.uniform(0.8, 0.95)
        nw, nh = int(W * r), int(H * r)
        left, top = random.randint(0, W - nw), random.randint(0, H - nh)
        out = im.crop((left, top, left + nw, top + nh)).resize((W, H), Image.LANCZOS)
    else:
        a = np.array(im)
        out = ImageOps.autocontrast(Image.fromarray(a + np.random.normal(0, 5, a.shape).astype(np.uint8)),
                                    cutoff=random.uniform(0, 2))
    out.save(dst, quality=95)
    return True


def _cpu_pool_job(src: Path, dst: Path, tasks, workers: int):
    """The WHOLE job the reference's way (dataset_balancer.py:70-81,97-168): the balanced tree starts as a copy of
    the originals (shutil.copytree), then one Pillow call chain per task in a pool of `workers` processes writes the
    generated files into it.  Returns (seconds for everything, seconds of the copy)."""
    import shutil
    from concurrent.futures import ProcessPoolExecutor
    t0 = time.perf_counter()
    shutil.copytree(src, dst)
    t_copy = time.perf_counter() - t0
    jobs = [(t["read_img"], str(dst / Path(t["output_path"]).relative_to(Path(t["output_path"]).parents[2])),
             t["transform_name"], t["seed"]) for t in tasks]
    with ProcessPoolExecutor(max_workers=workers) as ex:
        list(ex.map(_pil_task, jobs, chunksize=16))
    return time.perf_counter() - t0, t_copy


def augment_end_to_end(dev, generated: int = None):
    """The rate a user of `Augmentation.py` sees at BASELINE configs[2]'s size: DatasetBalancer.run() over a
    synthetic dataset of 72,000 224x224 JPEGs -> 100,000 generated files + manifest (LF_E2E_IMAGES sets the number
    of generated files; the copy of the originals, JPEG decode, H2D, kernels, D2H, JPEG encode, manifest all inside
    the timed region), and beside it THE SAME JOB run the reference's way — copytree, then one Pillow call chain
    per task in a process pool — with every granted core and with the reference's default worker count
    (dataset_balancer.py:41-44); the default-count run is bounded to a quarter of the tasks and scaled."""
    import shutil
    import tempfile

    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    from leaffliction_amd.utils.system_info import get_available_cores
    generated = int(os.environ.get("LF_E2E_IMAGES", "100000")) if generated is None else generated
    cores = min(get_available_cores(), usable_cores())
    tmp = Path(tempfile.mkdtemp(prefix="lf_e2e_"))
    cwd = os.getcwd()
    try:
        src, dst = tmp / "images", tmp / "augmented"
        n_orig = _e2e_make_dataset(src, dev, cores, _e2e_layout(generated))
        os.chdir(tmp)
        bal = DatasetBalancer(source_dir=str(src), target_dir=str(dst), seed=42, workers=cores)   # `--workers` = all granted cores
        bal.analyze_distribution()
        bal.calculate_plan()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bal.execute_balancing()
        sec = time.perf_counter() - t0
        out = {"images_per_sec": round(bal.completed / sec, 1), "generated": bal.completed, "failed": bal.failed,
               "originals": n_orig, "seconds": round(sec, 2), "codec_processes": bal.workers, "host_cores": cores,
               "stage_seconds": {k: round(v, 2) for k, v in bal.timings.items()},
               "pipeline_images_per_sec": round(bal.completed / max(
                   bal.timings.get("decode_kernels_encode", sec) - bal.timings.get("codec_pool_start", 0.0), 1e-6), 1),
               "includes": "copy of the originals, JPEG decode, H2D, kernels, D2H, JPEG encode (q=95), manifest"}
        tasks = bal.tasks
        shutil.rmtree(dst, ignore_errors=True)   # disk space for the CPU runs
        # the same job on host cores only: every task, all granted cores
        sec_all, copy_all = _cpu_pool_job(src, tmp / "cpu_all", tasks, cores)
        shutil.rmtree(tmp / "cpu_all", ignore_errors=True)
        pools = {"all_cores": {"images_per_sec": round(len(tasks) / sec_all, 1), "workers": cores, "tasks": len(tasks),
                               "seconds": round(sec_all, 2), "copy_seconds": round(copy_all, 2)}}
        # the reference's default worker count: the copy + every fourth task, the pool part scaled by four
        ref_default = max(1, int(cores * 0.75) // 2)   # dataset_balancer.py:41-44 with system_info.py:37-46
        part = tasks[::4]
        sec_def, copy_def = _cpu_pool_job(src, tmp / "cpu_def", part, ref_default)
        est = copy_def + (sec_def - copy_def) * (len(tasks) / max(1, len(part)))
        pools["reference_default_workers"] = {"images_per_sec": round(len(tasks) / est, 1), "workers": ref_default,
                                              "tasks": len(part), "seconds_scaled_to_all_tasks": round(est, 2),
                                              "copy_seconds": round(copy_def, 2)}
        out["cpu_pool_baseline"] = {"kind": "the reference's libraries on the same job (copytree, then the Pillow/numpy call "
                                            "chain of image_augmenter.py per task, one process per worker, file -> file)",
                                    "cores": cores, **pools}
        out["vs_cpu_all_cores"] = round(out["images_per_sec"] / pools["all_cores"]["images_per_sec"], 3)
        return out
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def inference_throughput(model, dev, batch=1024, iters=5):
    """Forward pass only (predict.py's batch mode, BASELINE configs[4] shape: 1024 images per GPU;
    fp32 here): uint8 batch resident in HBM -> pack/normalise -> conv stack (inference
    BatchNorm folded into the consumers' prologues) -> softmax -> argmax."""
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, 256, (batch, IMG, IMG, 3), dtype=torch.uint8, generator=g).to(dev)
    model.predict_device(x)   # warm-up at the timed batch size: the activation buffers are allocated here
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        labels = model.predict_device(x).argmax(-1)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / iters
    out = {"images_per_sec": round(batch / sec, 1), "batch": batch, "dtype": "f32",
           "tflops": round(TRAIN_GFLOP_PER_IMG / 3.0 * batch / sec / 1e3, 2),
           "labels_checksum": int(labels.sum().item())}
    # the reduced-precision mode of BASELINE configs[4]: bf16 conv operands, fp32 accumulation
    model.set_inference_dtype("bf16")
    model.predict_device(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        labels16 = model.predict_device(x).argmax(-1)
    torch.cuda.synchronize()
    sec16 = (time.perf_counter() - t0) / iters
    p16 = model.predict_device(x).clone()
    model.set_inference_dtype("f32")
    p32 = model.predict_device(x)
    top2 = p32.topk(2).values
    out["bf16"] = {"images_per_sec": round(batch / sec16, 1),
                   "tflops": round(TRAIN_GFLOP_PER_IMG / 3.0 * batch / sec16 / 1e3, 2),
                   # weights here are a few steps from random init: the fp32 top-2 margin is ~1e-3, far
                   # inside ANY reduced-precision error, so label agreement on THIS model says nothing; the
                   # confusion-matrix check on a trained model is tests/test_cnn_gpu.py
                   "max_abs_prob_diff_vs_f32": round(float((p16 - p32).abs().max().item()), 5),
                   "median_top2_margin_f32": round(float((top2[:, 0] - top2[:, 1]).median().item()), 5),
                   "labels_equal_to_f32": float((labels16 == labels).float().mean().item())}
    return out


def _fit_from_files(files, dev, batch=BATCH):
    """One epoch of `model.fit` over JPEG files, uncached (the reference's base preset, srcs/cli/train.py:38), on
    the mixed-precision step: the sequence's next batch decodes on the codec workers during the step
    (ManifestSequence.prefetch); beside it the plain host loader (one Pillow decode after the other) on a quarter
    of the files."""
    from leaffliction_amd.dataio.manifest import ManifestItem
    from leaffliction_amd.dataio.sequence import ManifestSequence
    from leaffliction_amd.model.cnn import build_leafcnn
    from leaffliction_amd.train.utils import build_loss, build_optimizer
    labels = sorted({Path(f).parent.name for f in files})
    l2i = {la: i for i, la in enumerate(labels)}
    items = [ManifestItem(str(i), "p", Path(f).parent.name, Path(f).parent.name, "train", Path(f))
             for i, f in enumerate(files)]
    cfg = {"optimizer": "adamw", "lr": 1e-3, "weight_decay": 1e-4, "label_smoothing": 0.02,
           "cosine_decay": False, "ema_decay": 0.0, "clipnorm": 0.5}

    def epoch(its, pooled):
        model, _ = build_leafcnn(num_classes=len(l2i), img_size=IMG, widths=list(WIDTHS), drop_block=0.15,
                                 drop_top=0.40, l2_reg=1e-4, seed=42)
        model.set_training_dtype("bf16")
        model.compile(build_optimizer(cfg, 1e-3), build_loss(cfg), ["accuracy"])
        mk = lambda part, shuffle: ManifestSequence(part, l2i, IMG, batch, shuffle, 42, num_classes=len(l2i),  # noqa: E731
                                                    one_hot=True)
        warm, seq = mk(its[:4 * batch], False), mk(its, True)
        for sq in (warm, seq):
            if not pooled:
                sq.prefetch = lambda idx: None
                sq.POOL_MIN = 10 ** 9
        model.fit(warm, epochs=1, verbose=0)               # graph capture, worker start-up
        seq._decoder, warm._decoder = warm._decoder, None   # the codec workers live across epochs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit(seq, epochs=1, verbose=0)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        seq.close()
        return len(its) / sec

    return {"images_per_sec": round(epoch(items, True), 1), "files": len(items), "batch": batch,
            "dtype": "bf16 step", "includes": "file reads, JPEG decode, resize, forward, backward, AdamW — one epoch "
                                              "of model.fit on an uncached ManifestSequence",
            "host_loader_images_per_sec": round(epoch(items[:max(4 * batch, len(items) // 4)], False), 1)}


def predict_end_to_end(dev, n_files=4096):
    """BASELINE configs[4] as a user of `predict -batch` sees it: JPEG files -> labels (Predictor.predict_batch: codec
    worker processes read the files' markers, the GPU does the decoding and the bf16 forward pass), next to the
    reference's way of feeding the same model — `ImageLoader.load_as_array` in a loop on one core (predictor.py's own
    loop, srcs/predict/predictor.py) — on a bounded sample."""
    import shutil
    import tempfile

    from leaffliction_amd.model.cnn import LeafCNN
    from leaffliction_amd.predict.predictor import Predictor
    tmp = Path(tempfile.mkdtemp(prefix="lf_pred_"))
    cwd = os.getcwd()
    pred = None
    try:
        src = tmp / "images"
        _e2e_make_dataset(src, dev, usable_cores(), _e2e_layout(11000))   # 7,920 originals
        files = sorted(str(p) for p in src.rglob("*.JPG"))[:n_files]
        os.chdir(tmp)
        model = LeafCNN(num_classes=NUM_CLASSES, img_size=IMG, widths=WIDTHS, drop_block=0.15, drop_top=0.40, l2_reg=1e-4,
                        augment=True, use_se=True, seed=42, device=dev)
        model.norm.mean[:] = 0.5
        model.norm.variance[:] = 1.0 / 12.0
        learn = tmp / "artifacts" / "models"
        learn.mkdir(parents=True)
        model.save(str(learn / "leaf_cnn.keras"))
        (learn / "meta.json").write_text(json.dumps({"model_file": str(learn / "leaf_cnn.keras"),
                                                     "labels": [f"class_{i}" for i in range(NUM_CLASSES)],
                                                     "data": {"img_size": IMG}}))
        os.environ["LEAFFLICTION_INFER_DTYPE"] = "bf16"
        pred = Predictor(learn)
        pred.load()
        pred.predict_batch(files[:256])            # warm-up: worker start-up, activation buffers
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pred.predict_batch(files)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        serial = files[:384]
        keep = Predictor.POOL_MIN
        Predictor.POOL_MIN = 10 ** 9
        try:
            t0 = time.perf_counter()
            ref = pred.predict_batch(serial)
            sec_ref = time.perf_counter() - t0
        finally:
            Predictor.POOL_MIN = keep
        same = all(a["top_prediction"] == b["top_prediction"] for a, b in zip(out, ref))
        # the training loader's side of the same step: filling ManifestSequence's HBM dataset cache (cache=True)
        from leaffliction_amd.dataio.manifest import ManifestItem
        from leaffliction_amd.dataio.sequence import ManifestSequence
        items = [ManifestItem(str(i), "p", "c", "c", "train", Path(f)) for i, f in enumerate(files)]
        t0 = time.perf_counter()
        seq = ManifestSequence(items, None, IMG, BATCH, False, 0, cache=True)
        torch.cuda.synchronize()
        sec_fill = time.perf_counter() - t0
        keep = ManifestSequence.POOL_MIN
        ManifestSequence.POOL_MIN = 10 ** 9
        try:
            t0 = time.perf_counter()
            seq_host = ManifestSequence(items[:384], None, IMG, BATCH, False, 0, cache=True)
            torch.cuda.synchronize()
            sec_fill_host = time.perf_counter() - t0
        finally:
            ManifestSequence.POOL_MIN = keep
        same_cache = bool(torch.equal(seq._cache_dev[:384], seq_host._cache_dev))
        del seq, seq_host
        fit_files = _fit_from_files(files, dev)
        return {"images_per_sec": round(len(out) / sec, 1), "files": len(out), "seconds": round(sec, 2),
                "dtype": "bf16", "includes": "file reads, JPEG decode (Huffman on host cores, the rest on the GPU), "
                                             "forward pass, per-file result records with the decoded pixels",
                "sequential_loop": {"images_per_sec": round(len(ref) / sec_ref, 1), "files": len(ref),
                                    "kind": "the reference's loop: one Pillow decode after the other, same model"},
                "same_labels_as_sequential_loop": bool(same),
                "loader_cache_fill": {"images_per_sec": round(len(items) / sec_fill, 1), "files": len(items),
                                      "includes": "codec worker start-up, file reads, JPEG decode, upload into the "
                                                  "HBM-resident uint8 dataset (ManifestSequence cache=True)",
                                      "host_loop_images_per_sec": round(384 / sec_fill_host, 1),
                                      "same_pixels_as_host_loop": same_cache},
                "fit_from_files": fit_files}
    finally:
        if pred is not None:
            pred.close()
        os.environ.pop("LEAFFLICTION_INFER_DTYPE", None)
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (weak scaling)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (the headline line, with the bf16 step as a second mode "
                         "inside it); bf16 = the mixed-precision step as the headline (configs[3] per-GPU work)")
    ap.add_argument("--no-bf16", action="store_true", help="f32 run: skip the second (bf16) training mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-augment", action="store_true", help="skip the augmentation-pass measurement")
    ap.add_argument("--no-inference", action="store_true", help="skip the forward-only measurement")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end Augmentation.py measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(--nproc-per-node N --master-addr 127.0.0.1)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # Rehearsal on a box with fewer GPUs than ranks (BENCH_BACKEND=gloo): ranks share the cards
    # and the collectives go through gloo; the driver's runs use one GPU per rank over RCCL.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    dp = None
    if world > 1:
        # the training path's own data-parallel glue (what cli.train uses): process group over RCCL, the flat
        # gradient bucket all-reduced in the dtype LEAFFLICTION_GRAD_BUCKET names
        from leaffliction_amd.train.parallel import DataParallel
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dp = DataParallel(backend=backend, device=dev)
        dist = dp.dist

    from leaffliction_amd import _lib
    from leaffliction_amd.model.cnn import LeafCNN
    _lib.load()
    timer = KernelTimer(_lib)
    timer.install()

    model = LeafCNN(num_classes=NUM_CLASSES, img_size=IMG, widths=WIDTHS, drop_block=0.15,
                    drop_top=0.40, l2_reg=1e-4, augment=True, use_se=True, seed=42, device=dev)
    model.norm.mean[:] = 0.5       # statistics of the synthetic uniform data
    model.norm.variance[:] = 1.0 / 12.0
    if world > 1:  # identical replicas
        dist.broadcast(model.flat_p, 0)
    gen = torch.Generator().manual_seed(42 + rank)
    n = args.batch
    x = torch.randint(0, 256, (n, IMG, IMG, 3), dtype=torch.uint8, generator=gen).to(dev)
    labels = torch.randint(0, NUM_CLASSES, (n,), generator=gen)
    y = (torch.nn.functional.one_hot(labels, NUM_CLASSES).float() * (1 - 0.02) + 0.02 / NUM_CLASSES).to(dev)

    total = args.warmup + args.steps
    # data-parallel: local gradients are scaled by 1/global_batch inside the step and the bucket is SUMmed over the
    # ranks by DataParallel.allreduce_grads (fp32 bucket, or bf16 under LEAFFLICTION_GRAD_BUCKET=bf16) — the same
    # call `fit` makes
    grad_sync = dp.allreduce_grads if dp is not None else None
    global_n = world * n if dp is not None else None
    step_kw = {"grad_sync": grad_sync, "global_n": global_n,
               # the bucket goes out in two pieces, the first beside the 224 x 224 layers' backward (train_step)
               "grad_overlap": dp if dp is not None and dp.overlap else None}

    def lr_at(step):
        return 2e-3 * 0.5 * (1.0 + math.cos(math.pi * min(step, total) / total))

    def timed_train(steps, warmup):
        """`warmup` untimed + exactly `steps` timed training steps between barrier + synchronize
        pairs; returns (seconds, max over ranks; final mean loss)."""
        nonlocal step
        for _ in range(warmup):
            model.train_step(x, y, lr_at(step), **step_kw)
            step += 1
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(steps):
            _probs, loss = model.train_step(x, y, lr_at(step), **step_kw)
            step += 1
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        sec = time.perf_counter() - t0
        timer.enabled = False
        if dist is not None:
            t = torch.tensor([sec], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec = float(t.item())
        fl = float(loss.mean())
        if not math.isfinite(fl):
            sys.exit(f"bench.py: non-finite loss {fl}")
        return sec, fl

    def bf16_report(sec, steps, fl, ksteps):
        """The mixed-precision step (BASELINE configs[3]'s per-GPU work): whole-step rate, and for
        the dominant kernel the HBM roofline (bf16 tensors make every convolution of this network
        bandwidth-bound: 72 FLOP per algorithmic byte at 224x224 / 32->32 against a ridge of ~310)."""
        kern = timer.summary()
        timer.records.clear()
        dname, dk = max(kern.items(), key=lambda kv: kv[1]["seconds"])
        conv_s = sum(v["seconds"] for v in kern.values())
        conv_b = sum(v["bytes"] for v in kern.values())
        conv_f = sum(v["flop"] for v in kern.values())
        gbs = dk["bytes"] / dk["seconds"] / 1e9
        traffic, traffic_source = None, None
        pmc = ROOT / "profiles" / "pmc_latest_bf16.json"
        if pmc.exists():   # committed rocprofv3 PMC summary of `bench.py --dtype bf16` (not taken in this run)
            try:
                doc = json.loads(pmc.read_text())
                tmpl = dname.split(">")[0].replace("<", "<").strip()
                # every instantiation of the dominant kernel's template with these leading arguments (its forward and
                # input-gradient variants), weighted by how often each was launched in the profiled command
                pre = tmpl.replace(" ", "").rstrip(">")
                hit = {kk: v for kk, v in doc.items() if not kk.startswith("_") and kk.replace(" ", "").startswith(pre)}
                cnt = doc.get("_launches", {})
                wsum = sum(cnt.get(kk, 1) for kk in hit)
                traffic = round(sum(v * cnt.get(kk, 1) for kk, v in hit.items()) / wsum) if hit else None
                traffic_source = {"file": "profiles/pmc_latest_bf16.json", **doc.get("_source", {})}
            except Exception:
                traffic = None
        return {"images_per_sec": round(world * n * steps / sec, 2), "ms_per_step": round(sec / steps * 1e3, 3),
                "steps": steps, "dtype": "bf16 storage + bf16 MFMA operands, fp32 accumulate / BN / SE / softmax / "
                                         "master weights",
                "step_tflops": round(TRAIN_GFLOP_PER_IMG * n * steps / sec / 1e3, 2),
                "frac_of_bf16_mfma_peak_2500TF": round(TRAIN_GFLOP_PER_IMG * n * steps / sec / 1e3 / 2500.0, 4),
                "final_loss": round(fl, 4),
                "roofline": {"bound": "hbm", "kernel": dname, "achieved": round(gbs, 1), "peak": 8000.0,
                             "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": traffic,
                             "traffic_unit": "L2 fabric-side bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE; "
                                             "includes Infinity-Cache hits)", "traffic_source": traffic_source,
                             "algorithmic_bytes_per_launch": round(dk["bytes"] / dk["launches"]),
                             "launches_per_step": dk["launches"] / ksteps,
                             "avg_launch_ms": round(dk["seconds"] / dk["launches"] * 1e3, 4),
                             "tflops": round(dk["flop"] / dk["seconds"] / 1e12, 1),
                             "measured": f"HIP events around each launch, {ksteps} eager steps right after the "
                                         "timed region (which replays the step as one HIP graph)"},
                "conv_all": {"GB_s": round(conv_b / conv_s / 1e9, 1), "tflops": round(conv_f / conv_s / 1e12, 1),
                             "ms_per_step": round(conv_s / ksteps * 1e3, 3)},
                "hip_graph": bool(model._graphs_on),
                "per_kernel": {k: {"ms": round(v["seconds"] / v["launches"] * 1e3, 4),
                                   "launches_per_step": v["launches"] / ksteps,
                                   "GB_s": round(v["bytes"] / v["seconds"] / 1e9, 1),
                                   "tflops": round(v["flop"] / v["seconds"] / 1e12, 1)}
                               for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["seconds"])}}

    def kernel_pass(steps=4):
        """Per-kernel durations for the roofline: HIP events around every convolution launch, on the
        stream it is launched on.  The timed region replays the step as ONE HIP graph (events cannot
        be recorded between its nodes), so the same step is run `steps` more times with eager
        launches right after it, same process, same buffers; the rocprofv3 summaries under profiles/
        are of the graph-free command as well."""
        nonlocal step
        timer.records.clear()
        graphs, model._graphs_on = model._graphs_on, False
        model.train_step(x, y, lr_at(step), **step_kw)   # eager warm-up
        torch.cuda.synchronize()
        timer.enabled = True
        for _ in range(steps):
            model.train_step(x, y, lr_at(step), **step_kw)
        torch.cuda.synchronize()
        timer.enabled = False
        model._graphs_on = graphs
        return steps

    step = 0
    bf16 = None
    kern_f32 = None
    if args.dtype == "bf16":
        model.set_training_dtype("bf16")
    elapsed, final_loss = timed_train(args.steps, args.warmup)
    timer.records.clear()
    ksteps = kernel_pass()
    if args.dtype == "bf16":
        bf16 = bf16_report(elapsed, args.steps, final_loss, ksteps)
    else:
        kern_f32 = timer.summary()
        timer.records.clear()
        if not args.no_bf16:
            # second mode: the same step on bf16 storage (its own warm-up: other kernels, other buffers)
            model.set_training_dtype("bf16")
            sec16, fl16 = timed_train(args.steps, min(args.warmup, 5))
            timer.records.clear()
            ksteps16 = kernel_pass()
            bf16 = bf16_report(sec16, args.steps, fl16, ksteps16)
            model.set_training_dtype("f32")
            timer.records.clear()

    if rank == 0 and args.dtype == "bf16":
        out = {"metric": "224x224 images/sec train", "value": bf16["images_per_sec"], "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": bf16["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic",
               "config": {"workload": "leaf_cnn base train step, mixed precision (configs[3] per-GPU work: img 224, "
                                      "batch 256/GPU, bf16, 8 classes, AdamW+clipnorm+EMA, in-model augmentation)",
                          "per_gpu_batch": n, "global_batch": world * n, "img_size": IMG,
                          "parallelism": f"dp{world}", "grad_bucket": dp.bucket_dtype if dp is not None else None,
                          "grad_overlap": bool(dp is not None and dp.overlap)},
               "roofline": bf16["roofline"], "conv_all": bf16["conv_all"], "step_tflops": bf16["step_tflops"],
               "final_loss": bf16["final_loss"], "per_kernel": bf16["per_kernel"]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    elif rank == 0:
        kern = kern_f32
        dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["seconds"])
        achieved = dom["flop"] / dom["seconds"] / 1e12
        conv_s = sum(v["seconds"] for v in kern.values())
        conv_f = sum(v["flop"] for v in kern.values())
        # HBM bytes per launch from rocprofv3 PMC passes are NOT taken in this run (counters need their
        # own passes): the figure is the committed summary's, with where it came from next to it
        traffic, traffic_source = None, None
        pmc = ROOT / "profiles" / "pmc_latest.json"
        if pmc.exists():
            try:
                doc = json.loads(pmc.read_text())
                traffic = doc.get(dom_name)
                traffic_source = {"file": "profiles/pmc_latest.json", **doc.get("_source", {"command": "scripts/profile_round.sh (round 1, tag j)"})}
            except Exception:
                traffic = None
        images = world * n * args.steps
        out = {
            "metric": "224x224 images/sec train",
            "value": round(images / elapsed, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "leaf_cnn base train step (configs[1]: img 224, batch 256/GPU, "
                                   "fp32, 8 classes, AdamW+clipnorm+EMA, in-model augmentation)",
                       "per_gpu_batch": n, "global_batch": world * n, "img_size": IMG,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": dom_name,
                         "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                         "launches_per_step": dom["launches"] / ksteps,
                         "measured": f"HIP events around each launch, {ksteps} eager steps right after the timed "
                                     "region (which replays the step as one HIP graph)",
                         "avg_launch_ms": round(dom["seconds"] / dom["launches"] * 1e3, 4),
                         "algorithmic_gflop_per_launch": round(dom["flop"] / dom["launches"] / 1e9, 3)},
            "conv_all": {"tflops": round(conv_f / conv_s / 1e12, 2),
                         "frac_of_mfma_peak": round(conv_f / conv_s / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                         "ms_per_step": round(conv_s / ksteps * 1e3, 3)},
            "hip_graph": bool(model._graphs_on),
            "step_tflops": round(TRAIN_GFLOP_PER_IMG * n * args.steps / elapsed / 1e3, 2),
            "final_loss": round(final_loss, 4),
        }
        if bf16 is not None:
            out["train_bf16"] = bf16
        if world == 1 and not args.no_inference:
            out["inference"] = inference_throughput(model, dev)
            if not args.no_e2e:
                out["inference"]["predict_end_to_end"] = predict_end_to_end(dev)
        if not args.no_augment and world == 1:
            del model
            torch.cuda.empty_cache()
            out["augment"] = augment_throughput(dev)
            out["augment"]["codec_edge"] = codec_edge_throughput(dev)
            if not args.no_cpu_baseline:
                out["augment"]["cpu_baseline"] = cpu_augment_baseline()
            if not args.no_e2e:
                out["augment"]["end_to_end"] = augment_end_to_end(dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dp is not None:
        dp.barrier()
        dp.shutdown()


if __name__ == "__main__":
    main()
