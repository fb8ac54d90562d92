"""Host-side parameter derivation for the geometric kernels.

Pillow derives rotate matrices, 16.16 fixed-point affine coefficients and resampling
tables in Python / C on the host before touching pixels; these helpers restate that
derivation (PIL/Image.py `rotate`, libImaging Geometry.c `affine_fixed`, Resample.c
`precompute_coeffs` + `normalize_coeffs_8bpc`) so the device kernels receive exactly the
numbers Pillow would use.  Reference call sites: srcs/preprocessing/image_augmenter.py:37
(rotate), :110 (crop + LANCZOS resize), srcs/utils/image_utils.py:109-114 (loader resize).
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0


def rotate_expand_matrix(w: int, h: int, angle_deg: float) -> Tuple[List[float], int, int]:
    """Inverse affine matrix and expanded canvas of `Image.rotate(angle, expand=True)`."""
    angle = angle_deg % 360.0
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0,
         round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]

    def tf(x: float, y: float) -> Tuple[float, float]:
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]

    m[2], m[5] = tf(-cx, -cy)
    m[2] += cx
    m[5] += cy
    pts = [tf(x, y) for x, y in ((0, 0), (w, 0), (w, h), (0, h))]
    nw = math.ceil(max(p[0] for p in pts)) - math.floor(min(p[0] for p in pts))
    nh = math.ceil(max(p[1] for p in pts)) - math.floor(min(p[1] for p in pts))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0)
    return m, nw, nh


def _fix16(v: float) -> int:
    t = v * 65536.0 + 0.5
    r = math.floor(t) if t < 0.0 else int(t)
    return ((int(r) + 2 ** 31) % 2 ** 32) - 2 ** 31  # C int wrap


def affine_fixed_coeffs(m: List[float]) -> List[int]:
    """Geometry.c affine_fixed: a0,a1,a3,a4 = FIX(a); a2,a5 include the half-pixel centre."""
    return [_fix16(m[0]), _fix16(m[1]), _fix16(m[2] + m[0] * 0.5 + m[1] * 0.5),
            _fix16(m[3]), _fix16(m[4]), _fix16(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x  # libm sin, as Resample.c (numpy's SIMD sin may differ by 1 ulp)


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


# (a crop of a W-pixel axis has ~W/8 widths x ~W/5 offsets: a balancing job asks for a few thousand distinct tables,
# ~1 ms of Python each; 512 entries thrashed)
@lru_cache(maxsize=8192)
def lanczos_coeffs(in_size: int, in0: float, in1: float, out_size: int):
    """(bounds int32 [out,2], coeffs int32 [out,ksize], ksize) for one axis."""
    scale = (in1 - in0) / out_size
    filterscale = scale if scale > 1.0 else 1.0
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ws = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in ws:
            ww += v
        for x, v in enumerate(ws):
            if ww != 0.0:
                v = v / ww
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx, 0] = xmin
        bounds[xx, 1] = xmax
    # lf_resample_u8 multiplies with 24-bit operands (normalised weights stay far inside)
    if int(np.abs(kk).max()) >= 1 << 23:
        raise ValueError("lanczos_coeffs: coefficient outside the 24-bit range of lf_resample_u8")
    return bounds, kk, ksize


def crop_resize_tables(w: int, h: int, left: int, top: int, nw: int, nh: int, ksize_pad: int = 0):
    """Tables for `img.crop((left,top,left+nw,top+nh)).resize((w,h), LANCZOS)`.

    The crop is folded into the bounds (window starts index the uncropped image).
    """
    xb, xk, kx = lanczos_coeffs(nw, 0.0, float(nw), w)
    yb, yk, ky = lanczos_coeffs(nh, 0.0, float(nh), h)
    xb = xb.copy()
    yb = yb.copy()
    xb[:, 0] += left
    yb[:, 0] += top
    return xb, xk, kx, yb, yk, ky


def pad_k(k: np.ndarray, ksize: int) -> np.ndarray:
    if k.shape[1] == ksize:
        return k
    out = np.zeros((k.shape[0], ksize), dtype=np.int32)
    out[:, :k.shape[1]] = k
    return out
