"""The host half of the JPEG encoder: libleafcodec.so (csrc/lf_jpeg_host.cpp built on its own, no HIP
runtime behind it) through ctypes.  It turns the quantised coefficients `ops.jpeg_fdct_quant_u8` leaves in
scan order into the complete file Pillow's `Image.save(path, quality=q)` writes for the same pixels
(srcs/utils/image_utils.py:49-56): markers, Annex K Huffman coding, byte stuffing.  Loaded by the codec
worker processes, which never touch the GPU."""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import numpy as np

LIB_PATH = Path(__file__).resolve().parent.parent / "libleafcodec.so"
_LIB: Optional[C.CDLL] = None


def load() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C leaffliction_amd/csrc`")
        lib = C.CDLL(str(LIB_PATH))
        lib.lf_jpeg_file_bound.argtypes = [C.c_int, C.c_int]
        lib.lf_jpeg_file_bound.restype = C.c_size_t
        lib.lf_jpeg_write_file.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.lf_jpeg_write_file.restype = C.c_long
        lib.lf_jpeg_wrap_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.lf_jpeg_wrap_scan.restype = C.c_long
        lib.lf_legacy_normal_u8.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.lf_legacy_normal_u8.restype = C.c_int
        lib.lf_jpeg_read_file.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.lf_jpeg_read_file.restype = C.c_int
        lib.lf_jpeg_scan_aux_offset.argtypes = [C.c_int, C.c_int]
        lib.lf_jpeg_scan_aux_offset.restype = C.c_size_t
        lib.lf_jpeg_scan_prepare.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                             C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        lib.lf_jpeg_scan_prepare.restype = C.c_int
        _LIB = lib
    return _LIB


_SCRATCH: Optional[np.ndarray] = None


def write_file(coef: np.ndarray, h: int, w: int, quality: int = 95) -> bytes:
    """coef: int16 [H/16 * W/16 MCUs, 6, 64] (any shape with that many elements, C-contiguous)."""
    global _SCRATCH
    lib = load()
    c = np.ascontiguousarray(coef, dtype=np.int16)
    if h <= 0 or w <= 0 or c.size != -(-h // 16) * -(-w // 16) * 384:
        raise ValueError(f"write_file: {c.size} coefficients do not make the MCUs of a {h}x{w} image")
    cap = int(lib.lf_jpeg_file_bound(h, w))
    if _SCRATCH is None or _SCRATCH.size < cap:
        _SCRATCH = np.empty(cap, dtype=np.uint8)
    n = int(lib.lf_jpeg_write_file(c.ctypes.data, h, w, int(quality), _SCRATCH.ctypes.data, cap))
    if n < 0:
        raise RuntimeError("lf_jpeg_write_file failed")
    return _SCRATCH[:n].tobytes()


QTAB_BYTES = 256   # two tables of 64 uint16 (luminance, chrominance), row-major


def read_file_into(data: bytes, dst: np.ndarray):
    """Parse + Huffman-decode a baseline 4:2:0 JPEG into `dst` (uint8 view of a slab slot): the two
    quantisation tables (256 bytes) followed by the quantised coefficients [MCUs, 6, 64] int16 in zigzag order.
    Returns (h, w), or None when the file is not of that kind (the caller decodes it with libjpeg) or does not fit."""
    lib = load()
    if dst.dtype != np.uint8 or not dst.flags["C_CONTIGUOUS"] or dst.size < QTAB_BYTES + 768:
        return None
    h, w = C.c_int(0), C.c_int(0)
    buf = np.frombuffer(data, dtype=np.uint8)
    base = dst.ctypes.data
    rc = lib.lf_jpeg_read_file(buf.ctypes.data, buf.size, base + QTAB_BYTES, (dst.size - QTAB_BYTES) // 2, base,
                               C.byref(h), C.byref(w))
    return (h.value, w.value) if rc == 0 else None


def scan_prepare_into(data: bytes, dst: np.ndarray):
    """Markers only: leave a baseline 4:2:0 file ready for the GPU's Huffman decoder (ops.jpeg_huffman_u8) in `dst`
    (uint8 view of a slab slot; layout in csrc/lf_jpeg_host.cpp).  Returns (h, w, hash of the file's Huffman tables,
    first byte of the slot behind what was written), or None when the file is to be decoded on the host
    (read_file_into, then libjpeg) or does not fit."""
    lib = load()
    if dst.dtype != np.uint8 or not dst.flags["C_CONTIGUOUS"]:
        return None
    h, w, hh = C.c_int(0), C.c_int(0), C.c_uint64(0)
    buf = np.frombuffer(data, dtype=np.uint8)
    rc = lib.lf_jpeg_scan_prepare(buf.ctypes.data, buf.size, dst.ctypes.data, dst.size, C.byref(h), C.byref(w), C.byref(hh))
    if rc != 0:
        return None
    aux = int(lib.lf_jpeg_scan_aux_offset(h.value, w.value))
    data_off, data_len = (int(v) for v in dst[aux + 24:aux + 32].view(np.uint32))
    return h.value, w.value, hh.value, aux + data_off + data_len + 16


def scan_aux_offset(h: int, w: int) -> int:
    return int(load().lf_jpeg_scan_aux_offset(int(h), int(w)))


def read_file(data: bytes):
    """(coef int16 [MCUs, 6, 64], qtab uint16 [2, 64], h, w) or None — convenience form of read_file_into."""
    slot = np.zeros(QTAB_BYTES + 3 * 4096 * 4096 // 16, dtype=np.uint8) if len(data) > (1 << 22) else \
        np.zeros(QTAB_BYTES + 3 * 1024 * 1024, dtype=np.uint8)
    hw = read_file_into(data, slot)
    if hw is None:
        return None
    h, w = hw
    n = (h // 16) * (w // 16)
    coef = slot[QTAB_BYTES:QTAB_BYTES + n * 768].view(np.int16).reshape(n, 6, 64).copy()
    return coef, slot[:QTAB_BYTES].view(np.uint16).reshape(2, 64).copy(), h, w


def wrap_scan(scan: np.ndarray, h: int, w: int, quality: int = 95) -> bytes:
    """Markers around a Huffman-coded scan made on the GPU (ops.jpeg_entropy_u8): the complete file."""
    lib = load()
    sc = np.ascontiguousarray(scan, dtype=np.uint8)
    out = np.empty(sc.size + 1024, dtype=np.uint8)
    n = int(lib.lf_jpeg_wrap_scan(sc.ctypes.data, sc.size, h, w, int(quality), out.ctypes.data, out.size))
    if n < 0:
        raise RuntimeError("lf_jpeg_wrap_scan failed")
    return out[:n].tobytes()


def legacy_normal_u8(seed: int, loc: float, scale: float, out: np.ndarray) -> None:
    """out[...] = np.random.RandomState(seed).normal(loc, scale, out.shape).astype(np.uint8), made in C (the same
    MT19937 stream, numpy's legacy polar Gaussian, libm's log and sqrt): `out` is a contiguous uint8 array."""
    if out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or not (0 <= int(seed) < 2 ** 32):
        raise ValueError("legacy_normal_u8: a contiguous uint8 array and a 32-bit seed")
    if load().lf_legacy_normal_u8(int(seed), float(loc), float(scale), out.size, out.ctypes.data, None) != 0:
        raise RuntimeError("lf_legacy_normal_u8 failed")


def legacy_normal_f64(seed: int, loc: float, scale: float, n: int) -> np.ndarray:
    """The float64 values behind legacy_normal_u8 (tests compare them with numpy's bit for bit)."""
    out = np.empty(int(n), dtype=np.float64)
    if load().lf_legacy_normal_u8(int(seed), float(loc), float(scale), out.size, None, out.ctypes.data) != 0:
        raise RuntimeError("lf_legacy_normal_u8 failed")
    return out
