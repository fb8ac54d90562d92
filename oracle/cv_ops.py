"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement (numpy) of the OpenCV 8-bit per-pixel arithmetic the reference's
`srcs/transform/filters` call on the hot path, plus `apply_mask`.

PARITY UNPINNED for the OpenCV functions: opencv-python (requirements.txt:10, unpinned) is
not installed here and cannot be installed (no network); the reference has no tests or
golden vectors for them.  The algorithms below restate OpenCV 4.x's published 8-bit paths
(imgproc color_hsv.simd.hpp RGB2HSV_b, color.simd_helpers RGB2Gray<uchar>, smooth.dispatch
GaussianBlurFixedPoint) and are guarded by known-answer tests in tests/test_oracle_kats.py
(HSV of cube corners and greys, gray coefficients, kernel sums, reflect-101 ramps).
`apply_mask` is pure numpy in the reference (srcs/utils/mask_utils.py:67-79) and its
docstring example is the pin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

HSV_SHIFT = 12


def _tables():
    sdiv = np.zeros(256, dtype=np.int64)
    hdiv = np.zeros(256, dtype=np.int64)
    for i in range(1, 256):
        # saturate_cast<int>(double) == cvRound: round half to even
        sdiv[i] = int(np.rint((255 << HSV_SHIFT) / (1.0 * i)))
        hdiv[i] = int(np.rint((180 << HSV_SHIFT) / (6.0 * i)))
    return sdiv, hdiv


_SDIV, _HDIV = _tables()


def rgb2hsv(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(rgb, cv2.COLOR_RGB2HSV) for uint8 (hist.py:184, blur.py:44); H in [0,180)."""
    r = rgb[..., 0].astype(np.int64)
    g = rgb[..., 1].astype(np.int64)
    b = rgb[..., 2].astype(np.int64)
    v = np.maximum(r, np.maximum(g, b))
    vmin = np.minimum(r, np.minimum(g, b))
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * _SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * _HDIV[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def rgb2gray(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(rgb, cv2.COLOR_RGB2GRAY) for uint8 (blur.py:27)."""
    r = rgb[..., 0].astype(np.int64)
    g = rgb[..., 1].astype(np.int64)
    b = rgb[..., 2].astype(np.int64)
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def gaussian_kernel_q8(ksize: int, sigma: float) -> np.ndarray:
    """getGaussianKernel + 8.8 fixed-point with error diffusion (taps sum to 256)."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    k = k / k.sum()
    q = [0] * ksize
    err = 0.0
    for i in range(ksize // 2):
        adj = k[i] * 256.0 + err
        v0 = int(np.floor(adj + 0.5))
        err = adj - v0
        q[i] = q[ksize - 1 - i] = v0
    q[ksize // 2] = 256 - 2 * sum(q[: ksize // 2])
    return np.asarray(q, dtype=np.int64)


def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(idx)
    idx = idx.copy()
    while ((idx < 0) | (idx >= n)).any():
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
    return idx


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(img, (ksize,ksize), sigma), uint8, BORDER_REFLECT_101 (blur.py:61,72).

    Fixed-point path: horizontal pass keeps Q8.8 in uint16, vertical pass rounds
    (acc + 2^15) >> 16.
    """
    q = gaussian_kernel_q8(ksize, sigma)
    r = ksize // 2
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[..., None]
    h, w, _ = src.shape
    xi = _reflect101(np.arange(-r, w + r), w)
    yi = _reflect101(np.arange(-r, h + r), h)
    padded = src[yi][:, xi]
    mid = np.zeros((h + 2 * r, w, src.shape[2]), dtype=np.int64)
    for k in range(ksize):
        mid += q[k] * padded[:, k:k + w]
    mid &= 0xFFFF
    acc = np.zeros((h, w, src.shape[2]), dtype=np.int64)
    for k in range(ksize):
        acc += q[k] * mid[k:k + h]
    out = ((acc + (1 << 15)) >> 16).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def apply_mask(img: np.ndarray, mask: np.ndarray, mask_color: str = "white") -> np.ndarray:
    """srcs/utils/mask_utils.py:44-83."""
    if mask_color.upper() == "WHITE":
        color_val = 255
    elif mask_color.upper() == "BLACK":
        color_val = 0
    else:
        raise ValueError(f'Mask Color {mask_color} is not "white" or "black"!')
    m = (mask > 127).astype(np.uint8) * 255
    out = img.copy()
    out[m == 0] = color_val
    return out


REGION_NAMES = ["leaf", "Vert Sain", "Vert Jaunâtre", "Jaune", "Brun/Orange", "Rouge",
                "Zones Sombres", "Zones Claires", "Violet/Pourpre", "hue Vert", "hue Jaune/Orange",
                "hue Rouge", "hue Violet", "hue Autres"]


def hsv_region_stats(rgb: np.ndarray):
    """Counters of apply_histogram_filter (hist.py:188, 38-65, 248-256) for one image.

    Returns (counts int64 [14], hsv_hist int64 [3,256]) over leaf = (s>10)&(v>15)&(v<245).
    """
    hsv = rgb2hsv(rgb)
    h = hsv[..., 0].astype(np.int64)
    s = hsv[..., 1].astype(np.int64)
    v = hsv[..., 2].astype(np.int64)
    leaf = (s > 10) & (v > 15) & (v < 245)
    preds = [
        leaf,
        leaf & (h >= 35) & (h <= 85) & (s >= 40) & (v >= 30),
        leaf & (h >= 20) & (h <= 40) & (s >= 25) & (v >= 30),
        leaf & (h >= 15) & (h <= 35) & (s >= 50) & (v >= 50),
        leaf & (((h >= 0) & (h <= 25)) | (h >= 160)) & (s >= 30) & (v >= 20),
        leaf & (((h >= 160) & (h <= 180)) | ((h >= 0) & (h <= 10))) & (s >= 40) & (v >= 30),
        leaf & (v <= 50) & (s >= 20),
        leaf & (v >= 200) & (s <= 30),
        leaf & (h >= 120) & (h <= 160) & (s >= 20),
        leaf & (h >= 35) & (h <= 85),
        leaf & (h >= 15) & (h <= 35),
        leaf & (((h >= 0) & (h <= 15)) | (h >= 160)),
        leaf & (h >= 120) & (h <= 160),
        leaf & (((h > 85) & (h < 120)) | ((h > 35) & (h < 15))),
    ]
    counts = np.array([int(p.sum()) for p in preds], dtype=np.int64)
    hist = np.stack([np.bincount(c[leaf], minlength=256) for c in (h, s, v)])
    return counts, hist
