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


# ---------------------------------------------------------------------------
# apply_blur_filter (srcs/transform/filters/blur.py:18-79).  PARITY UNPINNED like the rest of
# this file: the steps restate OpenCV 4.x's CPU paths (imgproc canny.cpp, deriv.cpp Sobel,
# morph.dispatch, core norm.cpp normalize + convert_scale.simd cvt_32f with its fused
# multiply-add) and numpy's float32 arithmetic in the order blur.py applies it.
# ---------------------------------------------------------------------------
def _replicate(idx: np.ndarray, n: int) -> np.ndarray:
    return np.clip(idx, 0, n - 1)


def sobel3(gray: np.ndarray, border: str):
    """cv2.Sobel(gray, ddepth, 1|0, 0|1, ksize=3) as exact integers (dx, dy).
    border: 'replicate' (what cv2.Canny asks for) or 'reflect101' (cv2.Sobel's default)."""
    h, w = gray.shape
    fn = _replicate if border == "replicate" else _reflect101
    yi = fn(np.arange(-1, h + 1), h)
    xi = fn(np.arange(-1, w + 1), w)
    p = gray.astype(np.int64)[yi][:, xi]
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return dx, dy


def canny(gray: np.ndarray, low: float, high: float, l2gradient: bool = True) -> np.ndarray:
    """cv2.Canny(gray, low, high, L2gradient=...), aperture 3.  L2gradient=True (blur.py:30): squared
    thresholds on dx^2 + dy^2; False, cv2's default (mask.py:792): floor(threshold) on |dx| + |dy|.
    22.5/67.5-degree sector tests in 15-bit fixed point (TG22 = 13573), asymmetric > / >= neighbour
    tests, 8-connected hysteresis."""
    h, w = gray.shape
    dx, dy = sobel3(gray, "replicate")
    mag = np.zeros((h + 2, w + 2), dtype=np.int64)
    if l2gradient:
        lo = int(np.floor(min(32767.0, low) ** 2)) if low > 0 else int(np.floor(low))
        hi = int(np.floor(min(32767.0, high) ** 2)) if high > 0 else int(np.floor(high))
        mag[1:-1, 1:-1] = dx * dx + dy * dy
    else:
        lo, hi = int(np.floor(low)), int(np.floor(high))
        mag[1:-1, 1:-1] = np.abs(dx) + np.abs(dy)
    m = mag[1:-1, 1:-1]
    x = np.abs(dx)
    y = np.abs(dy) << 15
    tg22x = x * 13573
    tg67x = tg22x + (x << 16)
    left, right = mag[1:-1, :-2], mag[1:-1, 2:]
    up, down = mag[:-2, 1:-1], mag[2:, 1:-1]
    s_pos = (dx ^ dy) < 0                      # s = +1 : previous row j-1, next row j+1
    prev_d = np.where(s_pos, mag[:-2, :-2], mag[:-2, 2:])
    next_d = np.where(s_pos, mag[2:, 2:], mag[2:, :-2])
    horiz = y < tg22x
    vert = ~horiz & (y > tg67x)
    diag = ~horiz & ~vert
    keep = (m > lo) & ((horiz & (m > left) & (m >= right)) | (vert & (m > up) & (m >= down)) |
                       (diag & (m > prev_d) & (m > next_d)))
    strong = keep & (m > hi)
    weak = keep & ~strong
    while True:
        pad = np.zeros((h + 2, w + 2), dtype=bool)
        pad[1:-1, 1:-1] = strong
        near = np.zeros((h, w), dtype=bool)
        for oy in range(3):
            for ox in range(3):
                near |= pad[oy:oy + h, ox:ox + w]
        grow = weak & near & ~strong
        if not grow.any():
            break
        strong |= grow
    return (strong * 255).astype(np.uint8)


def morph_cross3(img: np.ndarray, erode: bool) -> np.ndarray:
    """cv2.dilate / cv2.erode with getStructuringElement(MORPH_ELLIPSE, (3,3)) (a plus-shaped
    element), default border: pixels outside the image never win (blur.py:31-33,57-60)."""
    h, w = img.shape
    fill = 255 if erode else 0
    pad = np.full((h + 2, w + 2), fill, dtype=np.uint8)
    pad[1:-1, 1:-1] = img
    taps = [pad[1:-1, 1:-1], pad[:-2, 1:-1], pad[2:, 1:-1], pad[1:-1, :-2], pad[1:-1, 2:]]
    out = taps[0]
    for t in taps[1:]:
        out = np.minimum(out, t) if erode else np.maximum(out, t)
    return out


def normalize_minmax_f32(src: np.ndarray) -> np.ndarray:
    """cv2.normalize(src, None, 0, 255, cv2.NORM_MINMAX) for a float32 image: double scale and
    shift, then convertTo's float32 fused multiply-add per pixel."""
    src = src.astype(np.float32)
    smin, smax = float(src.min()), float(src.max())
    scale = 255.0 * ((1.0 / (smax - smin)) if (smax - smin) > np.finfo(np.float64).eps else 0.0)
    shift = 0.0 - smin * scale
    a, b = np.float32(scale), np.float32(shift)
    # fma(x, a, b) rounded once to float32: the 80-bit product/sum below is exact for these ranges
    wide = src.astype(np.longdouble) * np.longdouble(a) + np.longdouble(b)
    return wide.astype(np.float32)


def blur_saliency(rgb: np.ndarray, mask: np.ndarray, gaussian_sigma: float = 1.5,
                  brown_hue_range=(0, 30), brown_s_min: int = 20, brown_v_max: int = 200,
                  use_brown: bool = True) -> np.ndarray:
    """apply_blur_filter (blur.py:18-79) for one image given the leaf mask `make_mask` returned
    (defaults: srcs/transform/config.yaml:2,42-44)."""
    leaf = (mask > 0) if mask.ndim == 2 else (mask[..., 0] > 0)
    gray = rgb2gray(rgb)
    f32 = np.float32
    edges = morph_cross3(canny(gray, 50, 150), erode=False)
    sal = edges.astype(f32) * f32(0.4)
    gx, gy = sobel3(gray, "reflect101")
    gmag = np.sqrt((gx * gx + gy * gy).astype(f32))
    gnorm = normalize_minmax_f32(gmag).astype(np.uint8)
    sal = sal + gnorm.astype(f32) * f32(0.3)
    if use_brown:
        hsv = rgb2hsv(rgb)
        hh, ss, vv = hsv[..., 0], hsv[..., 1], hsv[..., 2]
        brown = ((hh >= brown_hue_range[0]) & (hh <= brown_hue_range[1]) & (ss >= brown_s_min) &
                 (vv <= brown_v_max) & leaf).astype(np.uint8) * 255
        closed = morph_cross3(morph_cross3(brown, erode=False), erode=True)
        grown = morph_cross3(morph_cross3(closed, erode=False), erode=False)
        sal = sal + grown.astype(f32) * f32(0.6)
    blurred = gaussian_blur(rgb, 15, 0.0)
    diff = np.abs(rgb.astype(f32) - blurred.astype(f32))
    cdiff = (diff[..., 0] + diff[..., 1] + diff[..., 2]) / f32(3.0)
    sal = sal + normalize_minmax_f32(cdiff) * f32(0.2)
    sal_u8 = normalize_minmax_f32(sal).astype(np.uint8)
    sal_blur = gaussian_blur(sal_u8, 5, gaussian_sigma)
    out = np.zeros_like(gray)
    out[leaf] = sal_blur[leaf]
    return np.repeat(out[..., None], 3, axis=2)


# ---------------------------------------------------------------------------
# _create_inclusive_mask (srcs/transform/filters/mask.py:727-831), the default strategy of make_mask.
# PARITY UNPINNED like the rest of this file (no cv2 here).  Restated: imgproc color_lab.cpp
# RGB2Lab_b (8-bit sRGB -> L*a*b*: gamma table at 3 fractional bits, 12-bit matrix, cube-root table
# at 15 bits), morph.dispatch (erode / dilate, border pixels never win), getStructuringElement's
# MORPH_ELLIPSE rows, connectedcomponents.cpp (8-connectivity, areas).  The two tables are built with
# double-precision pow / cbrt rounded to float32 where OpenCV uses its softfloat routines: a table
# entry that sits within one float32 ulp of a rounding boundary may differ by 1.
# ---------------------------------------------------------------------------
LAB_SHIFT, GAMMA_SHIFT = 12, 3
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
LAB_COEFFS = (1777, 1541, 778, 871, 2929, 296, 73, 448, 3575)   # round(4096 * sRGB->XYZ(D65) / white point)


def lab_tables():
    """(sRGBGammaTab_b [256], LabCbrtTab_b [3072]) of color_lab.cpp initLabTabs, uint16."""
    f32 = np.float32
    i = np.arange(256, dtype=np.float64)
    x = (i.astype(f32) / f32(255.0)).astype(np.float64)
    lin = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4).astype(f32)
    gamma = np.rint((f32(255 * (1 << GAMMA_SHIFT)) * lin).astype(np.float64)).astype(np.int64)
    j = np.arange(256 * 3 // 2 * (1 << GAMMA_SHIFT), dtype=np.float64)
    scale = f32(1.0) / (f32(255.0) * f32(1 << GAMMA_SHIFT))
    y = (scale * j.astype(f32)).astype(f32)
    lthresh, lscale, lbias = f32(216.0) / f32(24389.0), f32(841.0) / f32(108.0), f32(16.0) / f32(116.0)
    lowf = (y * lscale + lbias).astype(f32)     # mulAdd: float32 here (a fused form differs below 1e-7)
    fy = np.where(y < lthresh, lowf, np.cbrt(y.astype(np.float64)).astype(f32))
    cbrt = np.rint((f32(1 << LAB_SHIFT2) * fy).astype(np.float64)).astype(np.int64)
    return np.clip(gamma, 0, 65535).astype(np.uint16), np.clip(cbrt, 0, 65535).astype(np.uint16)


_LAB_GAMMA, _LAB_CBRT = lab_tables()


def rgb2lab(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(rgb, cv2.COLOR_RGB2LAB) for uint8 (mask.py:736): L in [0,255] (L* x 2.55), a and b + 128."""
    g = _LAB_GAMMA.astype(np.int64)
    r_, g_, b_ = g[rgb[..., 0]], g[rgb[..., 1]], g[rgb[..., 2]]
    c = LAB_COEFFS
    half = 1 << (LAB_SHIFT - 1)

    def f(c0, c1, c2):
        return _LAB_CBRT.astype(np.int64)[(r_ * c0 + g_ * c1 + b_ * c2 + half) >> LAB_SHIFT]

    fx, fy, fz = f(c[0], c[1], c[2]), f(c[3], c[4], c[5]), f(c[6], c[7], c[8])
    lscale = (116 * 255 + 50) // 100
    lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)
    half2 = 1 << (LAB_SHIFT2 - 1)
    L = (lscale * fy + lshift + half2) >> LAB_SHIFT2
    a = (500 * (fx - fy) + 128 * (1 << LAB_SHIFT2) + half2) >> LAB_SHIFT2
    b = (200 * (fy - fz) + 128 * (1 << LAB_SHIFT2) + half2) >> LAB_SHIFT2
    return np.clip(np.stack([L, a, b], -1), 0, 255).astype(np.uint8)


def ellipse_se(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) as a bool [k, k] array."""
    r = c = k // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    se = np.zeros((k, k), dtype=bool)
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))   # saturate_cast<int>: round half to even
            se[i, max(c - dx, 0):min(c + dx + 1, k)] = True
    return se


def morph(img: np.ndarray, se: np.ndarray, erode: bool) -> np.ndarray:
    """cv2.erode / cv2.dilate of a uint8 plane with a centred element, default border (outside never wins)."""
    h, w = img.shape
    kh, kw = se.shape
    ry, rx = kh // 2, kw // 2
    pad = np.full((h + 2 * ry, w + 2 * rx), 255 if erode else 0, dtype=np.uint8)
    pad[ry:ry + h, rx:rx + w] = img
    out = np.full((h, w), 255 if erode else 0, dtype=np.uint8)
    for i in range(kh):
        for j in range(kw):
            if se[i, j]:
                t = pad[i:i + h, j:j + w]
                out = np.minimum(out, t) if erode else np.maximum(out, t)
    return out


def morph_open(img: np.ndarray, k: int) -> np.ndarray:
    se = ellipse_se(k)
    return morph(morph(img, se, True), se, False)


def morph_close(img: np.ndarray, k: int) -> np.ndarray:
    se = ellipse_se(k)
    return morph(morph(img, se, False), se, True)


def largest_component(mask: np.ndarray) -> np.ndarray:
    """mask.py:817-824: keep the 8-connected component of the largest area (cv2.connectedComponentsWithStats;
    np.argmax takes the first of equal areas, and labels follow the raster order of each component's first
    pixel).  A plane with no foreground comes back unchanged."""
    h, w = mask.shape
    fg = mask > 0
    label = np.zeros((h, w), dtype=np.int64)
    areas = []
    for y0, x0 in zip(*np.nonzero(fg)):
        if label[y0, x0]:
            continue
        lab = len(areas) + 1
        stack, area = [(y0, x0)], 0
        label[y0, x0] = lab
        while stack:
            y, x = stack.pop()
            area += 1
            for yy in range(max(y - 1, 0), min(y + 2, h)):
                for xx in range(max(x - 1, 0), min(x + 2, w)):
                    if fg[yy, xx] and not label[yy, xx]:
                        label[yy, xx] = lab
                        stack.append((yy, xx))
        areas.append(area)
    if not areas:
        return mask.copy()
    best = 1 + int(np.argmax(areas))
    return ((label == best) * 255).astype(np.uint8)


def inclusive_mask(rgb: np.ndarray, green_hue_range=(25, 100)) -> np.ndarray:
    """_create_inclusive_mask (mask.py:727-831) on the working image: 0 / 255 plane
    (green_hue_range: srcs/transform/config.yaml:10)."""
    hsv, lab = rgb2hsv(rgb), rgb2lab(rgb)
    hh, ss, vv = (hsv[..., i].astype(np.int64) for i in range(3))
    ll, aa, bb = (lab[..., i].astype(np.int64) for i in range(3))
    # the reference compares uint8 planes: g > r + 15 is evaluated in uint8 and WRAPS (mask.py:759-763)
    r8, g8, b8 = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    lo, hi = green_hue_range
    lo, hi = max(0, lo - 10), min(179, hi + 15)
    strong_green = (hh >= lo) & (hh <= hi) & (ss >= 30) & (vv >= 30)
    u8 = np.uint8
    green_dominant = ((g8 > (r8 + u8(15))) | (g8 > (b8 + u8(15))) |
                      ((g8 > (r8 + u8(5))) & (g8 > (b8 + u8(5))) & (ss >= 20)))
    lab_green = (aa <= 125) & (bb >= 120) & (ll >= 20) & (ll <= 240)
    gray = rgb2gray(rgb)
    blur_gray = gaussian_blur(gray, 15, 0.0)
    texture = np.abs(gray.astype(np.int64) - blur_gray.astype(np.int64))
    background = (((ss <= 25) & (vv >= 50) & (vv <= 220)) |
                  ((hh >= 120) & (hh <= 160) & (ss >= 20) & (r8 > g8) & (b8 > g8)) |
                  ((ss <= 15) & (texture < 10)))
    edges = morph(canny(gray, 30, 100, l2gradient=False), ellipse_se(3), erode=False)
    plant = ((strong_green | green_dominant | lab_green | (edges > 0)) & ~background).astype(np.uint8) * 255
    plant = morph_open(plant, 3)
    plant = morph_close(plant, 9)
    plant = morph_close(plant, 7)
    plant = largest_component(plant)
    return morph_close(plant, 5)
