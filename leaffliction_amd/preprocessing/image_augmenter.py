"""GPU image augmenter with the reference's `ImageAugmenter` interface.

Mirror of srcs/preprocessing/image_augmenter.py:12-133: six file->file operations
`(image_path, output_path) -> bool`, dispatched by name, that never raise (log + False).
The random parameters are drawn from the process-global `random` / `np.random` streams
in exactly the reference's call order, so a given seed yields the reference's pixels;
the pixel work runs on the device through `leaffliction_amd.ops` (bit-exact with
Pillow, see tests/test_augment_gpu.py).  `draw_params` / `apply_batch` expose the same
ops for whole batches (used by DatasetBalancer's batched GPU path).
"""
from __future__ import annotations

import random
from typing import Any, Dict, List, Sequence

import numpy as np

from ..utils.common import get_logger
from ..utils.image_utils import ImageLoader

logger = get_logger(__name__)

TRANSFORMATIONS = ["flip", "rotate", "skew", "shear", "crop", "distortion"]
NOISE_LEVEL = 5


def draw_params(op: str, width: int, height: int, py_rng=random, np_rng=np.random) -> Dict[str, Any]:
    """Draw one op's random parameters with the reference's RNG calls, in its order.  By default
    from the process-global `random` / `np.random` streams (what the reference seeds per task);
    `random.Random(seed)` / `np.random.RandomState(seed)` instances produce the same values as
    seeding the globals with `seed`, and let independent tasks be drawn on separate threads."""
    if op == "flip":  # image_augmenter.py:23
        return {"mode": 0 if py_rng.choice([True, False]) else 1}
    if op == "rotate":  # :36
        return {"angle": py_rng.uniform(-30, 30)}
    if op == "skew":  # :48-59
        f = py_rng.uniform(0.05, 0.15)
        return {"coeffs": [1 + f, 0, -f * width, 0, 1 + f, -f * height, 0, 0]}
    if op == "shear":  # :77-82
        s = py_rng.uniform(-0.2, 0.2)
        if py_rng.choice([True, False]):
            return {"coeffs": [1, s, 0, 0, 1, 0, 0, 0]}
        return {"coeffs": [1, 0, 0, s, 1, 0, 0, 0]}
    if op == "crop":  # :101-107
        r = py_rng.uniform(0.8, 0.95)
        nw, nh = int(width * r), int(height * r)
        left = py_rng.randint(0, width - nw)
        top = py_rng.randint(0, height - nh)
        return {"box": (left, top, nw, nh)}
    if op == "distortion":  # :121, :127 — np.random first, then random.uniform
        noise = np_rng.normal(0, NOISE_LEVEL, (height, width, 3))
        return {"noise": noise, "cutoff": py_rng.uniform(0, 2)}
    raise AttributeError(op)


def apply_batch(op: str, x, params: Sequence[Dict[str, Any]], noise8=None) -> List:
    """Run `op` on a same-sized device batch x [N,H,W,3] u8.  Returns a list of N device
    tensors [h_i, w_i, 3] (rotate changes the size, the others keep it).  `noise8` (distortion): the batch's
    noise planes already on the device, uint8 [N,H,W,3] (ops.legacy_normal_u8)."""
    import torch

    from .. import ops
    dev = x.device
    if op == "flip":
        mode = torch.tensor([p["mode"] for p in params], dtype=torch.int32, device=dev)
        return list(ops.flip_u8(x, mode))
    if op == "rotate":
        return ops.rotate_expand_u8(x, [p["angle"] for p in params], fill=255)
    if op in ("skew", "shear"):
        co = torch.tensor([p["coeffs"] for p in params], dtype=torch.float64, device=dev)
        return list(ops.warp_bicubic_u8(x, co, perspective=(op == "skew"), axis_aligned=(op == "skew")))
    if op == "crop":
        return list(ops.crop_resize_lanczos_u8(x, [p["box"] for p in params]))
    if op == "distortion":
        cutoff = torch.tensor([p["cutoff"] for p in params], dtype=torch.float64, device=dev)
        if noise8 is not None:
            return list(ops.distortion_u8(x, cutoff, add=noise8))
        if "noise_seed" in params[0] and "noise8" not in params[0]:
            # the planes were left to the GPU by the codec workers and the caller has none: made here on the host
            from ..utils import jpeg_host
            for p in params:
                p["noise8"] = np.empty(tuple(x.shape[1:]), np.uint8)
                jpeg_host.legacy_normal_u8(int(p["noise_seed"]), 0.0, float(NOISE_LEVEL), p["noise8"])
        if "noise8" in params[0]:   # the codec workers cast the noise to uint8 (numpy's own astype)
            if all(isinstance(p["noise8"], torch.Tensor) for p in params):
                # views of the page-locked noise slab: one asynchronous copy per task, no stacking on the host
                n8 = torch.empty_like(x)
                for j, p in enumerate(params):
                    n8[j].copy_(p["noise8"], non_blocking=True)
            else:
                n8 = torch.from_numpy(np.stack([np.asarray(p["noise8"]) for p in params])).to(dev)
            return list(ops.distortion_u8(x, cutoff, add=n8))   # histogram taken while the noise is added
        noise = torch.from_numpy(np.stack([p["noise"] for p in params])).to(dev)
        return list(ops.autocontrast_u8(ops.noise_wrap_add_u8(x, noise), cutoff))
    raise AttributeError(op)


class ImageAugmenter:
    NOISE_LEVEL = NOISE_LEVEL

    def __init__(self, seed=None):
        if seed:  # seed 0 leaves the RNGs unseeded, as in the reference (:15-18)
            random.seed(seed)
            np.random.seed(seed)

    def _run(self, op: str, image_path, output_path) -> bool:
        try:
            import torch
            img = ImageLoader.load_as_array(image_path)
            h, w, _ = img.shape
            p = draw_params(op, w, h)
            x = torch.from_numpy(np.ascontiguousarray(img)).cuda().unsqueeze(0)
            out = apply_batch(op, x, [p])[0]
            ImageLoader.save_array(out.cpu().numpy(), output_path)
            return True
        except Exception as e:
            logger.error(f"Failed to process {image_path} - {e}")
            return False

    def flip(self, image_path, output_path):
        return self._run("flip", image_path, output_path)

    def rotate(self, image_path, output_path):
        return self._run("rotate", image_path, output_path)

    def skew(self, image_path, output_path):
        return self._run("skew", image_path, output_path)

    def shear(self, image_path, output_path):
        return self._run("shear", image_path, output_path)

    def crop(self, image_path, output_path):
        return self._run("crop", image_path, output_path)

    def distortion(self, image_path, output_path):
        return self._run("distortion", image_path, output_path)
