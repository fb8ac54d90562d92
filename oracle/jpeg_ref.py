"""ORACLE — test infrastructure only (never imported by the product path).

Baseline JPEG encoder restated from libjpeg(-turbo)'s integer pipeline, as Pillow drives it for
`Image.save(path, quality=95)` (the reference's `ImageLoader.save_pil_image`, srcs/utils/image_utils.py:49-56):
jccolor.c rgb_ycc_convert (16-bit fixed point), jcsample.c h2v2_downsample (alternating bias), jfdctint.c
(CONST_BITS 13, PASS1_BITS 2), jcdctmgr.c quantisation (round half away from zero on the 8x-scaled
coefficients), jchuff.c with the Annex K tables, jcmarker.c marker order.  4:2:0, baseline, no restart markers.

PINNED: Pillow (libjpeg-turbo) is installed here, so tests/test_jpeg_oracle.py compares `encode()` with the
BYTES Pillow writes, and every stage with the coefficients Pillow's own decoder reads back.
Ragged sizes follow libjpeg's padding (edge replication, dummy blocks); the DEcoder stages here cover whole MCUs only.
"""
from __future__ import annotations

import numpy as np

STD_LUM_Q = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64)
STD_CHR_Q = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.int64)
ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
    54, 47, 55, 62, 63], dtype=np.int64)   # jpeg_natural_order: zigzag position -> row-major index

DC_LUM_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
DC_LUM_VALS = list(range(12))
DC_CHR_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
DC_CHR_VALS = list(range(12))
AC_LUM_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d]
AC_LUM_VALS = [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14,
    0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09,
    0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a,
    0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65,
    0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88,
    0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9,
    0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea,
    0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]
AC_CHR_BITS = [0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77]
AC_CHR_VALS = [
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32,
    0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16,
    0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39,
    0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
    0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86,
    0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8,
    0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9,
    0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]


def quant_tables(quality: int):
    """jpeg_set_quality(quality, force_baseline=TRUE): (luminance, chrominance), row-major, 1..255."""
    q = max(1, min(100, int(quality)))
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = []
    for t in (STD_LUM_Q, STD_CHR_Q):
        out.append(np.clip((t * scale + 50) // 100, 1, 255))
    return out[0], out[1]


def _fix(x: float) -> int:
    return int(x * 65536 + 0.5)


def rgb_to_ycc(rgb: np.ndarray):
    """jccolor.c rgb_ycc_convert: three uint8 planes."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half, off = 1 << 15, 128 << 16
    y = (_fix(0.29900) * r + _fix(0.58700) * g + _fix(0.11400) * b + half) >> 16
    cb = (-_fix(0.16874) * r - _fix(0.33126) * g + _fix(0.50000) * b + off + half - 1) >> 16
    cr = (_fix(0.50000) * r - _fix(0.41869) * g - _fix(0.08131) * b + off + half - 1) >> 16
    return y.astype(np.uint8), cb.astype(np.uint8), cr.astype(np.uint8)


def h2v2_downsample(p: np.ndarray) -> np.ndarray:
    """jcsample.c h2v2_downsample: 2x2 box, bias 1, 2, 1, 2, ... along a row."""
    q = p.astype(np.int64)
    s = q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2]
    bias = np.tile(np.array([1, 2], dtype=np.int64), s.shape[1] // 2 + 1)[:s.shape[1]]
    return ((s + bias) >> 2).astype(np.uint8)


def fdct_islow(block: np.ndarray) -> np.ndarray:
    """jfdctint.c jpeg_fdct_islow on one 8x8 block of samples - 128 (int64 in, 8x-scaled coefficients out)."""
    CB, P1 = 13, 2
    F = {k: int(v * (1 << CB) + 0.5) for k, v in dict(
        a=0.298631336, b=0.390180644, c=0.541196100, d=0.765366865, e=0.899976223, f=1.175875602, g=1.501321110,
        h=1.847759065, i=1.961570560, j=2.053119869, k=2.562915447, l=3.072711026).items()}

    def descale(x, n):
        return (x + (1 << (n - 1))) >> n

    d = block.astype(np.int64).copy()
    for ps in (0, 1):
        if ps == 1:
            d = d.T.copy()
        t0, t7 = d[:, 0] + d[:, 7], d[:, 0] - d[:, 7]
        t1, t6 = d[:, 1] + d[:, 6], d[:, 1] - d[:, 6]
        t2, t5 = d[:, 2] + d[:, 5], d[:, 2] - d[:, 5]
        t3, t4 = d[:, 3] + d[:, 4], d[:, 3] - d[:, 4]
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        o = np.zeros_like(d)
        if ps == 0:
            o[:, 0] = (t10 + t11) << P1
            o[:, 4] = (t10 - t11) << P1
            sh = CB - P1
        else:
            o[:, 0] = descale(t10 + t11, P1)
            o[:, 4] = descale(t10 - t11, P1)
            sh = CB + P1
        z1 = (t12 + t13) * F["c"]
        o[:, 2] = descale(z1 + t13 * F["d"], sh)
        o[:, 6] = descale(z1 + t12 * (-F["h"]), sh)
        z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
        z5 = (z3 + z4) * F["f"]
        t4, t5, t6, t7 = t4 * F["a"], t5 * F["j"], t6 * F["l"], t7 * F["g"]
        z1, z2 = z1 * (-F["e"]), z2 * (-F["k"])
        z3, z4 = z3 * (-F["i"]) + z5, z4 * (-F["b"]) + z5
        o[:, 7] = descale(t4 + z1 + z3, sh)
        o[:, 5] = descale(t5 + z2 + z4, sh)
        o[:, 3] = descale(t6 + z2 + z3, sh)
        o[:, 1] = descale(t7 + z1 + z4, sh)
        d = o
    return d.T.copy()


def blocks_of(plane: np.ndarray) -> np.ndarray:
    """[H, W] -> [H/8, W/8, 8, 8]"""
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)


def quantised_coefficients(rgb: np.ndarray, quality: int = 95):
    """(Y [2*My, 2*Mx, 64], Cb [My, Mx, 64], Cr): quantised coefficients in ZIGZAG order, int16, for the
    My x Mx = ceil(H/16) x ceil(W/16) MCUs of the scan.  Ragged sizes as libjpeg pads them: samples are
    replicated to the right (full resolution, before the chroma box filter) and downwards (the last luminance row;
    the last full-resolution row up to an even count, then the last chroma row), and luminance blocks that lie
    wholly outside the image are dummies — no AC, the DC of the previous block of their MCU (jccoefct.c)."""
    h, w, _ = rgb.shape
    my, mx = -(-h // 16), -(-w // 16)
    hb, wb = -(-h // 8), -(-w // 8)          # luminance blocks that hold image samples
    ql, qc = quant_tables(quality)
    y, cb, cr = rgb_to_ycc(rgb)
    yi = np.minimum(np.arange(16 * my), h - 1)
    xi = np.minimum(np.arange(16 * mx), w - 1)
    ypad = y[yi][:, xi]
    # chroma: columns replicated at full resolution, rows to an even count, then the last CHROMA row repeated
    ch = -(-h // 2)
    crow = np.minimum(np.arange(8 * my), ch - 1)
    r0, r1 = 2 * crow, np.minimum(2 * crow + 1, h - 1)
    c0, c1 = np.minimum(2 * np.arange(8 * mx), w - 1), np.minimum(2 * np.arange(8 * mx) + 1, w - 1)
    bias = 1 + (np.arange(8 * mx) & 1)

    def chroma(p):
        q = p.astype(np.int64)
        s4 = q[r0][:, c0] + q[r0][:, c1] + q[r1][:, c0] + q[r1][:, c1]
        return ((s4 + bias) >> 2).astype(np.uint8)

    planes = (ypad, chroma(cb), chroma(cr))
    out = []
    for plane, q in zip(planes, (ql, qc, qc)):
        bl = blocks_of(plane.astype(np.int64) - 128)
        co = np.zeros(bl.shape[:2] + (64,), dtype=np.int64)
        div = (q << 3).astype(np.int64)
        for i in range(bl.shape[0]):
            for j in range(bl.shape[1]):
                c = fdct_islow(bl[i, j]).reshape(64)
                a = np.abs(c)
                v = (a + (div >> 1)) // div
                co[i, j] = (np.sign(c) * v)[ZIGZAG]
        out.append(co.astype(np.int16))
    yc = out[0]
    for i in range(my):          # dummy luminance blocks, MCU by MCU in buffer order Y00 Y01 Y10 Y11
        for j in range(mx):
            col1, row1 = 2 * j + 1 < wb, 2 * i + 1 < hb
            if not col1:
                yc[2 * i, 2 * j + 1] = 0
                yc[2 * i, 2 * j + 1, 0] = yc[2 * i, 2 * j, 0]
            if row1:
                if not col1:
                    yc[2 * i + 1, 2 * j + 1] = 0
                    yc[2 * i + 1, 2 * j + 1, 0] = yc[2 * i + 1, 2 * j, 0]
            else:
                yc[2 * i + 1, 2 * j] = 0
                yc[2 * i + 1, 2 * j + 1] = 0
                yc[2 * i + 1, 2 * j, 0] = yc[2 * i + 1, 2 * j + 1, 0] = yc[2 * i, 2 * j + 1, 0]
    return tuple(out)


def _huff_codes(bits, vals):
    code, k, table = 0, 0, {}
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


_DC = (_huff_codes(DC_LUM_BITS, DC_LUM_VALS), _huff_codes(DC_CHR_BITS, DC_CHR_VALS))
_AC = (_huff_codes(AC_LUM_BITS, AC_LUM_VALS), _huff_codes(AC_CHR_BITS, AC_CHR_VALS))


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, code: int, length: int):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _encode_block(bw: _Bits, zz, last_dc: int, tbl: int) -> int:
    diff = int(zz[0]) - last_dc
    a = -diff if diff < 0 else diff
    nb = a.bit_length()
    bw.put(*_DC[tbl][nb])
    if nb:
        bw.put(diff if diff >= 0 else diff - 1, nb)
    run = 0
    for k in range(1, 64):
        v = int(zz[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            bw.put(*_AC[tbl][0xF0])
            run -= 16
        a = -v if v < 0 else v
        nb = a.bit_length()
        bw.put(*_AC[tbl][(run << 4) | nb])
        bw.put(v if v >= 0 else v - 1, nb)
        run = 0
    if run:
        bw.put(*_AC[tbl][0x00])
    return int(zz[0])


def headers(h: int, w: int, quality: int = 95) -> bytes:
    ql, qc = quant_tables(quality)
    b = bytearray(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for i, q in enumerate((ql, qc)):
        b += b"\xff\xdb\x00\x43" + bytes([i]) + bytes(int(v) for v in q[ZIGZAG])
    b += b"\xff\xc0\x00\x11\x08" + bytes([h >> 8, h & 255, w >> 8, w & 255]) + b"\x03\x01\x22\x00\x02\x11\x01\x03\x11\x01"
    for tc_th, bits, vals in ((0x00, DC_LUM_BITS, DC_LUM_VALS), (0x10, AC_LUM_BITS, AC_LUM_VALS),
                              (0x01, DC_CHR_BITS, DC_CHR_VALS), (0x11, AC_CHR_BITS, AC_CHR_VALS)):
        n = 2 + 1 + 16 + len(vals)
        b += b"\xff\xc4" + bytes([n >> 8, n & 255, tc_th]) + bytes(bits) + bytes(vals)
    b += b"\xff\xda\x00\x0c\x03\x01\x00\x02\x11\x03\x11\x00\x3f\x00"
    return bytes(b)


def entropy_code(y, cb, cr) -> bytes:
    """Interleaved 4:2:0 scan of zigzag-ordered quantised coefficients -> stuffed, flushed entropy bytes."""
    bw = _Bits()
    last = [0, 0, 0]
    for my in range(cb.shape[0]):
        for mx in range(cb.shape[1]):
            for dy in (0, 1):
                for dx in (0, 1):
                    last[0] = _encode_block(bw, y[2 * my + dy, 2 * mx + dx], last[0], 0)
            last[1] = _encode_block(bw, cb[my, mx], last[1], 1)
            last[2] = _encode_block(bw, cr[my, mx], last[2], 1)
    bw.flush()
    return bytes(bw.out)


def encode(rgb: np.ndarray, quality: int = 95) -> bytes:
    """The file Pillow writes for Image.fromarray(rgb).save(path, quality=quality)."""
    h, w, _ = rgb.shape
    y, cb, cr = quantised_coefficients(rgb, quality)
    return headers(h, w, quality) + entropy_code(y, cb, cr) + b"\xff\xd9"


# ---------------------------------------------------------------------------
# Decoder side (libjpeg(-turbo) as Pillow drives it for Image.open(path).convert("RGB")): jidctint.c
# jpeg_idct_islow with the dequantisation folded in, jdsample.c h2v2_fancy_upsample (triangle filter, the
# default do_fancy_upsampling), jdcolor.c ycc_rgb_convert.  PINNED: tests compare with Pillow's decoded pixels.
# ---------------------------------------------------------------------------
def idct_islow(coef_rowmajor: np.ndarray, q_rowmajor: np.ndarray) -> np.ndarray:
    """One 8x8 block: quantised coefficients (row-major int) x quantisation table -> samples 0..255."""
    CB, P1 = 13, 2
    F = {k: int(v * (1 << CB) + 0.5) for k, v in dict(
        a=0.298631336, b=0.390180644, c=0.541196100, d=0.765366865, e=0.899976223, f=1.175875602, g=1.501321110,
        h=1.847759065, i=1.961570560, j=2.053119869, k=2.562915447, l=3.072711026).items()}
    d = (coef_rowmajor.astype(np.int64) * q_rowmajor.astype(np.int64)).reshape(8, 8)

    def one_pass(m, shift):   # 1-D IDCT along axis 0 of each column of m: m[k, col]
        z2, z3 = m[2], m[6]
        z1 = (z2 + z3) * F["c"]
        tmp2 = z1 + z3 * (-F["h"])
        tmp3 = z1 + z2 * F["d"]
        z2, z3 = m[0], m[4]
        tmp0 = (z2 + z3) << CB
        tmp1 = (z2 - z3) << CB
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        t0, t1, t2, t3 = m[7], m[5], m[3], m[1]
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F["f"]
        t0, t1, t2, t3 = t0 * F["a"], t1 * F["j"], t2 * F["l"], t3 * F["g"]
        z1, z2 = z1 * (-F["e"]), z2 * (-F["k"])
        z3, z4 = z3 * (-F["i"]) + z5, z4 * (-F["b"]) + z5
        t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
        half = 1 << (shift - 1)
        out = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
        return np.stack([(v + half) >> shift for v in out])

    ws = one_pass(d, CB - P1)                 # pass 1: columns (ws[row_out, col])
    res = one_pass(ws.T, CB + P1 + 3).T       # pass 2: rows
    return np.clip(res + 128, 0, 255).astype(np.uint8)


def h2v2_fancy_upsample(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: [h, w] -> [2h, 2w]; the image's first / last row and column stand in
    for their missing neighbours."""
    q = p.astype(np.int64)
    h, w = q.shape
    up = np.vstack([q[:1], q[:-1]])      # row above (row 0: itself)
    dn = np.vstack([q[1:], q[-1:]])      # row below (last row: itself)
    out = np.zeros((2 * h, 2 * w), dtype=np.int64)
    for v, nb in ((0, up), (1, dn)):
        cs = q * 3 + nb                                     # column sums, 16x scale after the horizontal step
        last = np.hstack([cs[:, :1], cs[:, :-1]])
        nxt = np.hstack([cs[:, 1:], cs[:, -1:]])
        out[v::2, 0::2] = (cs * 3 + last + 8) >> 4
        out[v::2, 1::2] = (cs * 3 + nxt + 7) >> 4
    return out.astype(np.uint8)


def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert (SCALEBITS 16)."""
    half = 1 << 15
    yy, xb, xr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = yy + ((_fix(1.40200) * xr + half) >> 16)
    b = yy + ((_fix(1.77200) * xb + half) >> 16)
    g = yy + ((-_fix(0.34414) * xb + half - _fix(0.71414) * xr) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode_coefficients(y, cb, cr, quality: int = 95) -> np.ndarray:
    """Zigzag-ordered quantised coefficients (as quantised_coefficients returns them) -> RGB [H, W, 3]."""
    ql, qc = quant_tables(quality)
    inv = np.argsort(ZIGZAG)   # row-major index -> zigzag position

    def plane(co, q):
        by, bx = co.shape[:2]
        out = np.zeros((by * 8, bx * 8), dtype=np.uint8)
        for i in range(by):
            for j in range(bx):
                out[8 * i:8 * i + 8, 8 * j:8 * j + 8] = idct_islow(co[i, j][inv], q)
        return out

    yp = plane(y, ql)
    return ycc_to_rgb(yp, h2v2_fancy_upsample(plane(cb, qc)), h2v2_fancy_upsample(plane(cr, qc)))
