"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement, in numpy / pure Python, of the arithmetic the reference's augmentation
loop performs through Pillow and numpy (the reference itself contains no arithmetic for
these ops: it calls `PIL.Image.transpose/rotate/transform/resize`, `ImageOps.autocontrast`
and `np.random.normal`, see the file:line cited per function).  Pillow is a third-party
dependency of the reference (requirements.txt:6, unpinned; 12.2.0 in this image), so the
published algorithm (libImaging Geometry.c / Resample.c / ImageOps.py) is restated here.

Pinning: tests/test_oracle_golden.py checks every function below against golden vectors
produced by running the reference's own `ImageAugmenter` / `ImageTransforms` in this
container (tests/golden/make_golden.py) — parity PINNED for this file.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import numpy as np

# ---------------------------------------------------------------------------
# flip — srcs/preprocessing/image_augmenter.py:20-31
# ---------------------------------------------------------------------------


def flip(img: np.ndarray, mode: int) -> np.ndarray:
    """mode 0 = Image.FLIP_LEFT_RIGHT, 1 = Image.FLIP_TOP_BOTTOM."""
    return img[:, ::-1].copy() if mode == 0 else img[::-1].copy()


# ---------------------------------------------------------------------------
# distortion — image_augmenter.py:116-133 (noise add with uint8 wrap + autocontrast)
# ---------------------------------------------------------------------------


def noise_wrap_add(img: np.ndarray, noise: np.ndarray) -> np.ndarray:
    """`np.clip(img + noise.astype(np.uint8), 0, 255)` — image_augmenter.py:121-124.

    float64 -> uint8 is a C cast: truncate toward zero then keep the low 8 bits (x86-64);
    uint8 + uint8 wraps; the clip is a no-op.
    """
    n8 = (np.trunc(noise).astype(np.int64) & 0xFF).astype(np.uint8)
    return (img.astype(np.uint16) + n8.astype(np.uint16)).astype(np.uint8)


def histogram(img: np.ndarray) -> np.ndarray:
    """PIL Image.histogram() of an RGB image: int64 [3,256]."""
    return np.stack([np.bincount(img[..., c].ravel(), minlength=256) for c in range(3)])


def autocontrast_lut(hist: np.ndarray, cutoff: float) -> np.ndarray:
    """PIL ImageOps.autocontrast's LUT (ImageOps.py, called at image_augmenter.py:127)."""
    lut = np.zeros((hist.shape[0], 256), dtype=np.uint8)
    for layer in range(hist.shape[0]):
        h = [int(v) for v in hist[layer]]
        if cutoff:
            n = sum(h)
            cut = int(n * cutoff // 100)
            for lo in range(256):
                if cut > h[lo]:
                    cut = cut - h[lo]
                    h[lo] = 0
                else:
                    h[lo] -= cut
                    cut = 0
                if cut <= 0:
                    break
            cut = int(n * cutoff // 100)
            for hi in range(255, -1, -1):
                if cut > h[hi]:
                    cut = cut - h[hi]
                    h[hi] = 0
                else:
                    h[hi] -= cut
                    cut = 0
                if cut <= 0:
                    break
        for lo in range(256):
            if h[lo]:
                break
        for hi in range(255, -1, -1):
            if h[hi]:
                break
        if hi <= lo:
            lut[layer] = np.arange(256)
        else:
            scale = 255.0 / (hi - lo)
            offset = -lo * scale
            for ix in range(256):
                v = int(ix * scale + offset)
                lut[layer, ix] = 0 if v < 0 else (255 if v > 255 else v)
    return lut


def lut_apply(img: np.ndarray, lut: np.ndarray) -> np.ndarray:
    """Image.point(lut) for RGB: per-channel table lookup."""
    out = np.empty_like(img)
    for c in range(3):
        out[..., c] = lut[c][img[..., c]]
    return out


def autocontrast(img: np.ndarray, cutoff: float) -> np.ndarray:
    return lut_apply(img, autocontrast_lut(histogram(img), cutoff))


# ---------------------------------------------------------------------------
# Image.transform(AFFINE | PERSPECTIVE, BICUBIC) — image_augmenter.py:44-94
# libImaging Geometry.c: affine_transform / perspective_transform + bicubic_filter32RGB
# ---------------------------------------------------------------------------


def _bicubic(v1, v2, v3, v4, d):
    p1 = v2
    p2 = -v1 + v3
    p3 = 2 * (v1 - v2) + v3 - v4
    p4 = -v1 + v2 - v3 + v4
    return p1 + d * (p2 + d * (p3 + d * p4))


def warp_bicubic(img: np.ndarray, coeffs, perspective: bool) -> np.ndarray:
    """Output size == input size; pixels that map outside the source are black (fill=1)."""
    h, w, _ = img.shape
    a = [float(c) for c in coeffs] + [0.0] * (8 - len(coeffs))
    ys, xs = np.mgrid[0:h, 0:w]
    xin = xs + 0.5
    yin = ys + 0.5
    if perspective:
        den = a[6] * xin + a[7] * yin + 1
        sx = (a[0] * xin + a[1] * yin + a[2]) / den
        sy = (a[3] * xin + a[4] * yin + a[5]) / den
    else:
        sx = a[0] * xin + a[1] * yin + a[2]
        sy = a[3] * xin + a[4] * yin + a[5]
    inside = ~((sx < 0.0) | (sx >= w) | (sy < 0.0) | (sy >= h))
    sx = np.where(inside, sx, 0.5) - 0.5
    sy = np.where(inside, sy, 0.5) - 0.5
    # FLOOR(v): (int)floor(v) for v < 0, (int)v otherwise == floor for all v
    x = np.floor(sx).astype(np.int64)
    y = np.floor(sy).astype(np.int64)
    dx = sx - x
    dy = sy - y
    x -= 1
    y -= 1
    xc = [np.clip(x + k, 0, w - 1) for k in range(4)]
    src = img.astype(np.float64)
    out = np.zeros_like(img)
    for b in range(3):
        plane = src[..., b]
        rows = []
        for k in range(4):
            yk = y + k
            if k == 0:
                yy = np.clip(yk, 0, h - 1)
                rows.append(_bicubic(plane[yy, xc[0]], plane[yy, xc[1]], plane[yy, xc[2]],
                                     plane[yy, xc[3]], dx))
            else:
                ok = (yk >= 0) & (yk < h)
                yy = np.clip(yk, 0, h - 1)
                v = _bicubic(plane[yy, xc[0]], plane[yy, xc[1]], plane[yy, xc[2]],
                             plane[yy, xc[3]], dx)
                rows.append(np.where(ok, v, rows[k - 1]))
        v = _bicubic(rows[0], rows[1], rows[2], rows[3], dy)
        # bicubic_filter32RGB: (UINT8)v1 — truncation, no +0.5 (pinned by the golden vectors)
        q = np.where(v <= 0.0, 0, np.where(v >= 255.0, 255, v.astype(np.int64)))
        out[..., b] = np.where(inside, q, 0).astype(np.uint8)
    return out


def skew_coeffs(width: int, height: int, f: float):
    """image_augmenter.py:50-59."""
    return [1 + f, 0, -f * width, 0, 1 + f, -f * height, 0, 0]


def shear_coeffs(s: float, horizontal: bool):
    """image_augmenter.py:79-82."""
    return [1, s, 0, 0, 1, 0] if horizontal else [1, 0, 0, s, 1, 0]


# ---------------------------------------------------------------------------
# Image.rotate(angle, expand=True, fillcolor="white") — image_augmenter.py:33-42
# PIL/Image.py rotate() matrix + libImaging Geometry.c affine_fixed (NEAREST, 16.16)
# ---------------------------------------------------------------------------


def rotate_matrix(w: int, h: int, angle: float):
    """Returns (matrix6, out_w, out_h) exactly as PIL.Image.Image.rotate(expand=True)."""
    angle = angle % 360.0
    if angle in (0, 180, 90, 270):
        raise NotImplementedError("transpose fast paths are not on the augmenter's path")
    cx, cy = w / 2, h / 2
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0,
         round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]

    def tf(x, y, mm):
        a, b, c, d, e, f = mm
        return a * x + b * y + c, d * x + e * y + f

    m[2], m[5] = tf(-cx, -cy, m)
    m[2] += cx
    m[5] += cy
    xx, yy = [], []
    for x, y in ((0, 0), (w, 0), (w, h), (0, h)):
        tx, ty = tf(x, y, m)
        xx.append(tx)
        yy.append(ty)
    nw = math.ceil(max(xx)) - math.floor(min(xx))
    nh = math.ceil(max(yy)) - math.floor(min(yy))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0, m)
    return m, nw, nh


def _fix(v: float) -> int:
    """Geometry.c FIX(v) = FLOOR(v*65536.0 + 0.5), wrapped to int32."""
    t = v * 65536.0 + 0.5
    r = int(math.floor(t)) if t < 0.0 else int(t)
    return ((r + 2**31) % 2**32) - 2**31


def affine_fixed_coeffs(m):
    a0, a1, a3, a4 = _fix(m[0]), _fix(m[1]), _fix(m[3]), _fix(m[4])
    a2 = _fix(m[2] + m[0] * 0.5 + m[1] * 0.5)
    a5 = _fix(m[5] + m[3] * 0.5 + m[4] * 0.5)
    return [a0, a1, a2, a3, a4, a5]


def affine_nearest_fixed(img: np.ndarray, fix6, out_w: int, out_h: int, fill: int) -> np.ndarray:
    h, w, _ = img.shape
    a0, a1, a2, a3, a4, a5 = [np.int64(v) for v in fix6]
    ys, xs = np.mgrid[0:out_h, 0:out_w].astype(np.int64)

    def wrap32(v):
        return ((v + 2**31) % 2**32) - 2**31

    xx = wrap32(a2 + a1 * ys + a0 * xs)
    yy = wrap32(a5 + a4 * ys + a3 * xs)
    xin = xx >> 16
    yin = yy >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.full((out_h, out_w, 3), fill, dtype=np.uint8)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def rotate_expand_white(img: np.ndarray, angle: float) -> np.ndarray:
    h, w, _ = img.shape
    m, nw, nh = rotate_matrix(w, h, angle)
    return affine_nearest_fixed(img, affine_fixed_coeffs(m), nw, nh, 255)


# ---------------------------------------------------------------------------
# Image.resize(size, LANCZOS) — image_utils.py:109-114, image_augmenter.py:110
# libImaging Resample.c: precompute_coeffs + normalize_coeffs_8bpc + two 8-bit passes
# ---------------------------------------------------------------------------

PRECISION_BITS = 32 - 8 - 2


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int, support: float = 3.0,
                      filt=_lanczos):
    """Returns (bounds int32 [out,2] (xmin, count), kk int32 [out,ksize], ksize)."""
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(
                0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        k = kk[xx, :cnt].astype(np.int64).reshape((cnt,) + (1,) * (src.ndim - 1))
        acc = (1 << (PRECISION_BITS - 1)) + (src[xmin:xmin + cnt] * k).sum(axis=0)
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_lanczos(img: np.ndarray, out_w: int, out_h: int, box=None) -> np.ndarray:
    """PIL `img.resize((out_w,out_h), LANCZOS, box)`; box = (x0,y0,x1,y1) floats or None.

    Horizontal pass first, then vertical (Resample.c ImagingResample); a pass is skipped
    when that axis is unchanged; same size and full box is a plain copy (Image.resize).
    """
    h, w, _ = img.shape
    if box is None:
        box = (0, 0, w, h)
    if (out_w, out_h) == (w, h) and tuple(box) == (0, 0, w, h):
        return img.copy()
    need_h = out_w != w or box[0] != 0 or box[2] != w
    need_v = out_h != h or box[1] != 0 or box[3] != h
    cur = img
    if need_v:
        yb, yk, _ = precompute_coeffs(h, box[1], box[3], out_h)
    if need_h:
        xb, xk, _ = precompute_coeffs(w, box[0], box[2], out_w)
        if need_v:  # only the rows the vertical pass reads are produced
            first = int(yb[0, 0])
            last = int(yb[-1, 0] + yb[-1, 1])
            tmp = _resample_axis(cur[first:last], xb, xk, 1)
            yb = yb.copy()
            yb[:, 0] -= first
            cur = tmp
        else:
            cur = _resample_axis(cur, xb, xk, 1)
    if need_v:
        cur = _resample_axis(cur, yb, yk, 0)
    return cur


def crop_resize_lanczos(img: np.ndarray, left: int, top: int, nw: int, nh: int) -> np.ndarray:
    """image_augmenter.py:109-110: img.crop(box).resize((W,H), LANCZOS)."""
    h, w, _ = img.shape
    return resize_lanczos(img[top:top + nh, left:left + nw], w, h)


# ---------------------------------------------------------------------------
# pack / normalize — image_utils.py:117-130
# ---------------------------------------------------------------------------


def normalize_array(arr: np.ndarray) -> np.ndarray:
    return arr.astype(np.float32) / 255.0


def pack_nchw(batch_u8: np.ndarray, mean=None, denom=None) -> np.ndarray:
    """[N,H,W,3] u8 -> [N,3,H,W] f32 (x/255, then (v-mean)/denom if given)."""
    x = normalize_array(batch_u8)
    if mean is not None:
        x = (x - np.asarray(mean, np.float32)) / np.asarray(denom, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))
