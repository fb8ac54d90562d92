"""Host-side launchers for the augmentation kernels of libleafhip.so.

torch supplies device memory and the current HIP stream; every function checks dtype,
contiguity and device on the host before handing raw pointers to the C ABI, so a kernel
is never launched with operand shapes other than the ones its grid assumes.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

_U8, _I32, _F32, _F64 = torch.uint8, torch.int32, torch.float32, torch.float64


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str, ndim: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.LeafHipError(f"{name}: expected a CUDA(HIP) tensor — there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    return t


def _hwc(t: torch.Tensor, name: str):
    _chk(t, _U8, name, 4)
    n, h, w, c = t.shape
    if c != 3 or n == 0:
        raise ValueError(f"{name}: expected [N,H,W,3] with N>0, got {tuple(t.shape)}")
    return n, h, w


def pack_hwc_u8_to_nchw_f32(x: torch.Tensor, mean: Optional[Sequence[float]] = None,
                            denom: Optional[Sequence[float]] = None,
                            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N,H,W,3] u8 -> [N,3,H,W] f32 = x/255 (then (v-mean)/denom per channel if given)."""
    n, h, w = _hwc(x, "pack.x")
    if out is None:
        out = torch.empty((n, 3, h, w), dtype=_F32, device=x.device)
    _chk(out, _F32, "pack.out", 4)
    if tuple(out.shape) != (n, 3, h, w):
        raise ValueError("pack.out: shape mismatch")
    m = d = None
    if mean is not None:
        m = (_lib.c_float * 3)(*[float(v) for v in mean])
        d = (_lib.c_float * 3)(*[float(v) for v in denom])
    _lib.call("lf_pack_hwc_u8_to_nchw_f32", x.data_ptr(), out.data_ptr(), n, h, w, m, d, _stream())
    return out


def hist_u8(x: torch.Tensor) -> torch.Tensor:
    """Per-image per-channel 256-bin histogram, int32 [N,3,256]."""
    n, h, w = _hwc(x, "hist.x")
    hist = torch.empty((n, 3, 256), dtype=_I32, device=x.device)
    _lib.call("lf_hist_u8", x.data_ptr(), hist.data_ptr(), n, h, w, _stream())
    return hist


def autocontrast_lut(hist: torch.Tensor, cutoff: torch.Tensor) -> torch.Tensor:
    _chk(hist, _I32, "autocontrast_lut.hist", 3)
    _chk(cutoff, _F64, "autocontrast_lut.cutoff", 1)
    n = hist.shape[0]
    if tuple(hist.shape) != (n, 3, 256) or cutoff.shape[0] != n:
        raise ValueError("autocontrast_lut: hist must be [N,3,256] and cutoff [N]")
    lut = torch.empty((n, 3, 256), dtype=_U8, device=hist.device)
    _lib.call("lf_autocontrast_lut", hist.data_ptr(), cutoff.data_ptr(), lut.data_ptr(), n,
              _stream())
    return lut


def lut_apply_u8(x: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    n, h, w = _hwc(x, "lut_apply.x")
    _chk(lut, _U8, "lut_apply.lut", 3)
    if tuple(lut.shape) != (n, 3, 256):
        raise ValueError("lut_apply.lut: expected [N,3,256]")
    out = torch.empty_like(x)
    _lib.call("lf_lut_apply_u8", x.data_ptr(), lut.data_ptr(), out.data_ptr(), n, h, w, _stream())
    return out


def autocontrast_u8(x: torch.Tensor, cutoff: torch.Tensor) -> torch.Tensor:
    """PIL ImageOps.autocontrast(img, cutoff) for a batch: hist -> LUT -> point."""
    return lut_apply_u8(x, autocontrast_lut(hist_u8(x), cutoff))


def gather_images_u8(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """Batch = src[index] for a device-resident uint8 dataset [M,H,W,3]; index int32 [B] (device)."""
    m, h, w = _hwc(src, "gather.src")
    _chk(index, _I32, "gather.index", 1)
    b = index.shape[0]
    if b == 0:
        raise ValueError("gather.index: empty batch")
    out = torch.empty((b, h, w, 3), dtype=_U8, device=src.device)
    for k in range(0, b, 65535):
        kk = min(b, k + 65535)
        _lib.call("lf_gather_rows_u8", src.data_ptr(), index[k:kk].data_ptr(), out[k:kk].data_ptr(),
                  kk - k, h * w * 3, _stream())
    return out


def flip_u8(x: torch.Tensor, mode: torch.Tensor) -> torch.Tensor:
    """mode[n] = 0: FLIP_LEFT_RIGHT, 1: FLIP_TOP_BOTTOM."""
    n, h, w = _hwc(x, "flip.x")
    _chk(mode, _I32, "flip.mode", 1)
    if mode.shape[0] != n:
        raise ValueError("flip.mode: expected [N]")
    out = torch.empty_like(x)
    _lib.call("lf_flip_u8", x.data_ptr(), out.data_ptr(), mode.data_ptr(), n, h, w, _stream())
    return out


def noise_wrap_add_u8(x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    _chk(x, _U8, "noise.x")
    _chk(noise, _F64, "noise.noise")
    if noise.shape != x.shape or x.numel() == 0:
        raise ValueError("noise: noise must have the image's shape")
    out = torch.empty_like(x)
    _lib.call("lf_noise_wrap_add_u8", x.data_ptr(), noise.data_ptr(), out.data_ptr(), x.numel(),
              _stream())
    return out


def add_wrap_u8(x: torch.Tensor, add: torch.Tensor) -> torch.Tensor:
    """x + add (mod 256), bytewise: the distortion's noise add when the noise is uint8 already."""
    _chk(x, _U8, "add_wrap.x")
    _chk(add, _U8, "add_wrap.add")
    if add.shape != x.shape or x.numel() == 0 or x.numel() % 4:
        raise ValueError("add_wrap: same non-empty shape, size a multiple of 4 bytes")
    out = torch.empty_like(x)
    _lib.call("lf_add_wrap_u8", x.data_ptr(), add.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def noise_philox_add_u8(x: torch.Tensor, seed: int, sigma: float = 5.0) -> torch.Tensor:
    _chk(x, _U8, "noise_philox.x")
    if x.numel() == 0:
        raise ValueError("noise_philox: empty input")
    out = torch.empty_like(x)
    _lib.call("lf_noise_philox_add_u8", x.data_ptr(), out.data_ptr(), x.numel(),
              int(seed) & (2**64 - 1), float(sigma), _stream())
    return out


def noise_hist_u8(x: torch.Tensor, add: Optional[torch.Tensor] = None, seed: int = 0, sigma: float = 5.0):
    """(x + noise mod 256, its per-channel histograms [N,3,256]) in one pass over the batch: `add` is the uint8 noise
    plane (add_wrap_u8's), or None for the device-drawn Philox noise of noise_philox_add_u8 (same seed, same bytes).
    Falls back to the two separate kernels when an image is not a multiple of 16 bytes."""
    n, h, w = _hwc(x, "noise_hist.x")
    if add is not None:
        _chk(add, _U8, "noise_hist.add")
        if add.shape != x.shape:
            raise ValueError("noise_hist.add: same shape as the batch")
    if (h * w * 3) % 16 or x.data_ptr() % 16 or (add is not None and add.data_ptr() % 16):
        y = add_wrap_u8(x, add) if add is not None else noise_philox_add_u8(x, seed, sigma)
        return y, hist_u8(y)
    out = torch.empty_like(x)
    hist = torch.empty((n, 3, 256), dtype=_I32, device=x.device)
    _lib.call("lf_noise_hist_u8", x.data_ptr(), None if add is None else add.data_ptr(), out.data_ptr(),
              hist.data_ptr(), n, h, w, int(seed) & (2**64 - 1), float(sigma), _stream())
    return out, hist


def legacy_normal_u8(seeds: Sequence[int], loc: float, scale: float, count: int, device) -> tuple:
    """np.random.RandomState(seed).normal(loc, scale, count).astype(np.uint8) for every seed (0 <= seed < 2**32), made on
    the GPU: (planes uint8 [N, count], flags int32 [N]).  flags[i] != 0: plane i may differ from numpy's in a byte (a
    value within 1e-9 of an integer, where the last bit of log() decides the cast) — make it with
    utils.jpeg_host.legacy_normal_u8 instead (about one 224 x 224 x 3 plane in 3,000)."""
    if not len(seeds) or any(not 0 <= int(v) < 2 ** 32 for v in seeds) or count <= 0:
        raise ValueError("legacy_normal: seeds must be in [0, 2**32) and count positive")
    n = len(seeds)
    sd = torch.from_numpy(np.asarray(seeds, dtype=np.uint32).view(np.int32)).to(device)
    stride = (int(count) + 15) // 16 * 16
    out = torch.empty((n, stride), dtype=_U8, device=device)
    flags = torch.empty(n, dtype=_I32, device=device)
    _lib.call("lf_legacy_normal_batch_u8", sd.data_ptr(), float(loc), float(scale), int(count), out.data_ptr(), stride, n,
              flags.data_ptr(), _stream())
    return out[:, :count], flags


def distortion_u8(x: torch.Tensor, cutoff: torch.Tensor, add: Optional[torch.Tensor] = None, seed: int = 0,
                  sigma: float = 5.0) -> torch.Tensor:
    """ImageAugmenter.distortion on a batch (image_augmenter.py:121-131): noise add (+ histogram in the same pass),
    autocontrast LUT, LUT apply — four image passes."""
    y, hist = noise_hist_u8(x, add, seed, sigma)
    return lut_apply_u8(y, autocontrast_lut(hist, cutoff))


def mask_composite_u8(img: torch.Tensor, mask: torch.Tensor, mask_color: str = "white"):
    """apply_mask: out = mask > 127 ? img : (255 if white else 0)."""
    if mask_color.upper() == "WHITE":
        color = 255
    elif mask_color.upper() == "BLACK":
        color = 0
    else:
        raise ValueError(f'Mask Color {mask_color} is not "white" or "black"!')
    n, h, w = _hwc(img, "mask_composite.img")
    _chk(mask, _U8, "mask_composite.mask", 3)
    if tuple(mask.shape) != (n, h, w):
        raise ValueError("mask_composite.mask: expected [N,H,W]")
    out = torch.empty_like(img)
    _lib.call("lf_mask_composite_u8", img.data_ptr(), mask.data_ptr(), out.data_ptr(), n, h, w,
              color, _stream())
    return out


def rgb2hsv_u8(x: torch.Tensor) -> torch.Tensor:
    n, h, w = _hwc(x, "rgb2hsv.x")
    out = torch.empty_like(x)
    _lib.call("lf_rgb2hsv_u8", x.data_ptr(), out.data_ptr(), n * h * w, _stream())
    return out


def rgb2gray_u8(x: torch.Tensor) -> torch.Tensor:
    n, h, w = _hwc(x, "rgb2gray.x")
    out = torch.empty((n, h, w), dtype=_U8, device=x.device)
    _lib.call("lf_rgb2gray_u8", x.data_ptr(), out.data_ptr(), n * h * w, _stream())
    return out


def hsv_region_stats(x: torch.Tensor):
    """Returns (counts int32 [N,14], hsv_hist int32 [N,3,256]) — see include/leafhip.h."""
    n, h, w = _hwc(x, "hsv_region_stats.x")
    counts = torch.empty((n, 14), dtype=_I32, device=x.device)
    hh = torch.empty((n, 3, 256), dtype=_I32, device=x.device)
    _lib.call("lf_hsv_region_stats", x.data_ptr(), counts.data_ptr(), hh.data_ptr(), n, h, w,
              _stream())
    return counts, hh


def gaussian_kernel_q8(ksize: int, sigma: float) -> np.ndarray:
    """OpenCV getGaussianKernel + 8.8 fixed-point quantisation (sum == 256).

    sigma <= 0 follows cv2: 0.3*((ksize-1)*0.5 - 1) + 0.8.  The fixed-point conversion
    rounds each tap and carries the rounding error forward so that the taps sum to 256
    (OpenCV's getGaussianKernelFixedPoint_ED).  Parity unpinned: cv2 is not installable here.
    """
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    xs = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(xs * xs) / (2.0 * sigma * sigma))
    k /= k.sum()
    q = np.zeros(ksize, dtype=np.int64)
    err = 0.0
    half = ksize // 2
    # symmetric error diffusion from the edges towards the centre
    for i in range(half):
        v = k[i] * 256.0 + err
        q[i] = q[ksize - 1 - i] = int(np.floor(v + 0.5))
        err = v - q[i]
    q[half] = 256 - 2 * int(q[:half].sum())
    return q.astype(np.uint16)


def gauss_blur_u8(x: torch.Tensor, ksize: int, sigma: float) -> torch.Tensor:
    """cv2.GaussianBlur(x, (ksize, ksize), sigma) for [N,H,W,3] or [N,H,W] uint8."""
    _chk(x, _U8, "gauss_blur.x")
    if x.dim() == 4 and x.shape[-1] == 3:
        n, h, w, ch = x.shape
    elif x.dim() == 3:
        (n, h, w), ch = x.shape, 1
    else:
        raise ValueError("gauss_blur.x: expected [N,H,W,3] or [N,H,W]")
    kq = np.ascontiguousarray(gaussian_kernel_q8(ksize, sigma).astype(np.uint16))  # host constants
    out = torch.empty_like(x)
    _lib.call("lf_gauss_blur_u8", x.data_ptr(), out.data_ptr(), n, h, w, ch, kq.ctypes.data, ksize,
              _stream())
    return out


def blur_saliency_u8(x: torch.Tensor, leaf_mask: torch.Tensor, gaussian_sigma: float = 1.5,
                     brown_hue_range=(0, 30), brown_s_min: int = 20, brown_v_max: int = 200,
                     use_brown: bool = True) -> torch.Tensor:
    """apply_blur_filter (srcs/transform/filters/blur.py:18-79) for a batch [N,H,W,3] uint8 and
    the leaf masks [N,H,W] uint8 (leaf = mask > 0) its make_mask_func produced; defaults are
    srcs/transform/config.yaml:2,42-44.  Returns the gray saliency image replicated to RGB."""
    n, h, w = _hwc(x, "blur_saliency.x")
    _chk(leaf_mask, _U8, "blur_saliency.leaf_mask", 3)
    if tuple(leaf_mask.shape) != (n, h, w) or leaf_mask.device != x.device:
        raise ValueError(f"blur_saliency.leaf_mask: expected {[n, h, w]} on {x.device}, got "
                         f"{list(leaf_mask.shape)} on {leaf_mask.device}")
    kq15 = np.ascontiguousarray(gaussian_kernel_q8(15, 0.0).astype(np.uint16))  # host constants
    kq5 = np.ascontiguousarray(gaussian_kernel_q8(5, float(gaussian_sigma)).astype(np.uint16))
    nbytes = int(_lib.load().lf_blur_saliency_workspace(n, h, w))
    ws = torch.empty(nbytes, dtype=_U8, device=x.device)
    out = torch.empty_like(x)
    _lib.call("lf_blur_saliency_u8", x.data_ptr(), leaf_mask.data_ptr(), out.data_ptr(), n, h, w,
              1 if use_brown else 0, int(brown_hue_range[0]), int(brown_hue_range[1]),
              int(brown_s_min), int(brown_v_max), kq15.ctypes.data, kq5.ctypes.data, ws.data_ptr(),
              nbytes, _stream())
    return out


def inclusive_mask_u8(x: torch.Tensor, green_hue_range=(25, 100)) -> torch.Tensor:
    """_create_inclusive_mask (srcs/transform/filters/mask.py:727-831) for a batch [N,H,W,3] uint8 of working
    images: the leaf mask [N,H,W] uint8 (0 / 255).  green_hue_range: srcs/transform/config.yaml:10."""
    n, h, w = _hwc(x, "inclusive_mask.x")
    kq15 = np.ascontiguousarray(gaussian_kernel_q8(15, 0.0).astype(np.uint16))  # host constants
    nbytes = int(_lib.load().lf_inclusive_mask_workspace(n, h, w))
    ws = torch.empty(nbytes, dtype=_U8, device=x.device)
    out = torch.empty((n, h, w), dtype=_U8, device=x.device)
    _lib.call("lf_inclusive_mask_u8", x.data_ptr(), out.data_ptr(), n, h, w, int(green_hue_range[0]),
              int(green_hue_range[1]), kq15.ctypes.data, ws.data_ptr(), nbytes, _stream())
    return out


def jpeg_fdct_quant_u8(x: torch.Tensor, quality: int = 95, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The pixel half of Image.save(path, quality=quality) (image_utils.py:49-56) for a batch [N,H,W,3] uint8
    : libjpeg's quantised DCT coefficients, int16 [N, ceil(H/16) * ceil(W/16), 6, 64] — per MCU the
    blocks Y00 Y01 Y10 Y11 Cb Cr in zigzag order, what utils.jpeg_host.write_file turns into the file."""
    n, h, w = _hwc(x, "jpeg_fdct_quant.x")
    shape = (n, -(-h // 16) * -(-w // 16), 6, 64)   # ragged sizes carry libjpeg's padding (replicated edges, dummy blocks)
    if out is None:
        out = torch.empty(shape, dtype=torch.int16, device=x.device)
    elif out.dtype != torch.int16 or out.numel() != n * shape[1] * 384 or not out.is_contiguous():
        raise ValueError("jpeg_fdct_quant.out: expected a contiguous int16 tensor of N*MCUs*384 elements")
    _lib.call("lf_jpeg_fdct_quant_u8", x.data_ptr(), out.data_ptr(), n, h, w, int(quality), _stream())
    return out.view(shape)


def jpeg_entropy_u8(coef: torch.Tensor, h: int, w: int, out: Optional[torch.Tensor] = None,
                    out_stride: Optional[int] = None) -> torch.Tensor:
    """The Huffman coding of Image.save (Annex K tables, byte stuffing) for the coefficients jpeg_fdct_quant_u8
    made: int16 [N, MCUs, 6, 64] -> uint8 [N, out_stride], row = int32 length (-1: did not fit) then the scan;
    utils.jpeg_host.wrap_scan puts the markers around it."""
    n = coef.shape[0]
    mcus = -(-h // 16) * -(-w // 16)
    if coef.dtype != torch.int16 or not coef.is_contiguous() or coef.numel() != n * mcus * 384:
        raise ValueError("jpeg_entropy.coef: expected the contiguous int16 output of jpeg_fdct_quant_u8")
    stride = int(out_stride or (4 + mcus * 768 + 4095) // 4096 * 4096)
    if out is None:
        out = torch.empty((n, stride), dtype=_U8, device=coef.device)
    elif out.dtype != _U8 or tuple(out.shape) != (n, stride) or not out.is_contiguous():
        raise ValueError("jpeg_entropy.out: expected a contiguous uint8 [N, out_stride] tensor")
    nbytes = int(_lib.load().lf_jpeg_entropy_workspace(n, stride))
    ws = torch.empty(nbytes, dtype=_U8, device=coef.device)
    _lib.call("lf_jpeg_entropy_u8", coef.data_ptr(), mcus * 768, out.data_ptr(), stride, n, h, w, ws.data_ptr(),
              nbytes, _stream())
    return out


def jpeg_encode_items_u8(buf: torch.Tensor, items: Sequence[Sequence[int]], room: int, quality: int = 95) -> None:
    """Image.save(path, quality=q) up to the markers for images of DIFFERENT sizes lying in one flat uint8 device buffer
    (the balancer's rotated canvases in their slots of the output mirror), two launches for all of them: items[i] =
    (byte offset of image i's pixels, h, w); the finished scan (int32 length, then the bytes; -1 when it does not fit
    `room` bytes) REPLACES the pixels at the same offset (a multiple of 4).  utils.jpeg_host.wrap_scan puts the
    markers around each scan: the complete file, Pillow's bytes."""
    _chk(buf, _U8, "jpeg_encode_items.buf", 1)
    n = len(items)
    if n == 0:
        return
    lib = _lib.load()
    desc = np.zeros((n, 6), dtype=np.int64)   # lf_jpeg_item: four int64, then h, w | nblocks, reserved as int32 pairs
    hw = desc[:, 4:].view(np.int32)
    groups = coef = 0
    for i, (off, h, w) in enumerate(items):
        off, h, w = int(off), int(h), int(w)
        if off % 4 or off < 0 or h <= 0 or w <= 0 or off + max(h * w * 3, 4) > buf.numel() or off + room > buf.numel():
            raise ValueError(f"jpeg_encode_items: image {i} ({h}x{w} at {off}) does not lie in the buffer")
        mcus = -(-h // 16) * -(-w // 16)
        desc[i, :4] = (off, coef, off, groups)
        hw[i] = (h, w, mcus * 6, 0)
        groups += int(lib.lf_jpeg_fdct_groups(h, w))
        coef += mcus * 384
    if room < 1024 or room % 4:
        raise ValueError("jpeg_encode_items: room must be a multiple of 4 and at least 1 KiB")
    d = torch.from_numpy(desc).to(buf.device)
    cws = torch.empty(coef, dtype=torch.int16, device=buf.device)
    nbytes = int(lib.lf_jpeg_entropy_workspace(n, room))
    ews = torch.empty(nbytes, dtype=_U8, device=buf.device)
    _lib.call("lf_jpeg_fdct_quant_items_u8", buf.data_ptr(), cws.data_ptr(), d.data_ptr(), n, groups, int(quality), _stream())
    _lib.call("lf_jpeg_entropy_items_u8", cws.data_ptr(), d.data_ptr(), buf.data_ptr(), int(room), n, ews.data_ptr(), nbytes,
              _stream())


def jpeg_huffman_u8(slots: torch.Tensor, h: int, w: int, sequential: bool = False) -> torch.Tensor:
    """The Huffman step of Image.open for N files of one size that utils.jpeg_host.scan_prepare_into has laid into
    `slots` (uint8 [N, slot_bytes], contiguous): each row's coefficient area [256, 256 + 3hw) is written IN PLACE
    (then jpeg_idct_rgb_u8 as for host-decoded files).  One workgroup per image decodes 256 pieces of the scan at
    once (the restart intervals of a file that has them: one each); scans of a megabyte and more — or all files, with
    `sequential` — go through the
    one-lane-per-image kernel, which wants one set of Huffman tables per 64 consecutive rows.
    Returns int32 [N] on the device: 0 decoded, 1 malformed / truncated scan (hand the file to libjpeg), 2 tables
    differ from the group's (one-lane-per-image kernel), 3 no prepared scan in the slot."""
    _chk(slots, _U8, "jpeg_huffman.slots", 2)
    n, stride = slots.shape
    if h % 16 or w % 16 or stride % 16 or slots.stride(0) != stride or slots.data_ptr() % 16:
        raise ValueError(f"jpeg_huffman: {h}x{w} images need contiguous 16-byte aligned rows (multiple of 16 bytes)")
    status = torch.empty(n, dtype=torch.int32, device=slots.device)
    _lib.call("lf_jpeg_huffman_u8", slots.data_ptr(), stride, n, h, w, status.data_ptr(), 1 if sequential else 0, _stream())
    return status


def jpeg_idct_rgb_u8(slots: torch.Tensor, h: int, w: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The pixel half of Image.open(path).convert("RGB") (image_utils.py:19-33) for N images of one size whose
    files utils.jpeg_host.read_file_into has Huffman-decoded: `slots` uint8 [N, slot_bytes], each row = the
    file's two quantisation tables (256 bytes) then its quantised coefficients (3*h*w bytes).  Returns RGB
    [N, h, w, 3] uint8 — Pillow's pixels, bit for bit."""
    _chk(slots, _U8, "jpeg_idct_rgb.slots", 2)
    n, stride = slots.shape
    if h % 16 or w % 16 or stride < 256 + 3 * h * w or stride % 16 or slots.stride(0) != stride:
        raise ValueError(f"jpeg_idct_rgb: {h}x{w} images need contiguous rows of >= {256 + 3 * h * w} bytes (multiple of 16)")
    if out is None:
        out = torch.empty((n, h, w, 3), dtype=_U8, device=slots.device)
    nbytes = int(_lib.load().lf_jpeg_decode_workspace(n, h, w))
    ws = torch.empty(nbytes, dtype=_U8, device=slots.device)
    _lib.call("lf_jpeg_idct_rgb_u8", slots.data_ptr() + 256, stride, slots.data_ptr(), stride, out.data_ptr(),
              n, h, w, ws.data_ptr(), nbytes, _stream())
    return out


# ---------------------------------------------------------------------------
# geometric ops (Pillow semantics)
# ---------------------------------------------------------------------------
from .preprocessing import geometry as _geo  # noqa: E402


def warp_bicubic_u8(x: torch.Tensor, coeffs: torch.Tensor, perspective: bool,
                    axis_aligned: bool = False) -> torch.Tensor:
    """Image.transform(size, AFFINE|PERSPECTIVE, coeffs, BICUBIC); coeffs f64 [N,8].
    axis_aligned: speed hint for pure scale maps (a1 = a3 = 0); verified on the device."""
    n, h, w = _hwc(x, "warp_bicubic.x")
    _chk(coeffs, _F64, "warp_bicubic.coeffs", 2)
    if tuple(coeffs.shape) != (n, 8):
        raise ValueError("warp_bicubic.coeffs: expected [N,8] float64")
    out = torch.empty_like(x)
    _lib.call("lf_warp_bicubic_u8", x.data_ptr(), out.data_ptr(), coeffs.data_ptr(),
              (1 if perspective else 0) | (2 if axis_aligned else 0), n, h, w, _stream())
    return out


def rotate_expand_plan(w: int, h: int, angles: Sequence[float], device, offsets: Optional[Sequence[int]] = None,
                       limit: Optional[int] = None):
    """Host-side part of Image.rotate(expand=True): fixed-point coefficients, canvas sizes and
    output offsets for a batch (device tensors) — reusable across launches.  Offsets are packed (16-byte
    aligned) unless the caller names them (`offsets`, 16-byte aligned, e.g. the slots of an output slab; with
    `limit` the plan is None when a canvas does not fit its `limit` bytes)."""
    fix, ohw, offs, off = [], [], [], 0
    for i, a in enumerate(angles):
        m, nw, nh = _geo.rotate_expand_matrix(w, h, float(a))
        fix.append(_geo.affine_fixed_coeffs(m))
        ohw.append((nh, nw))
        if offsets is not None:
            if int(offsets[i]) % 16 or (limit is not None and nh * nw * 3 > limit):
                return None
            offs.append(int(offsets[i]))
            off = max(off, int(offsets[i]) + ((nh * nw * 3 + 15) // 16) * 16)
            continue
        offs.append(off)
        off += ((nh * nw * 3 + 15) // 16) * 16
    return {"fix": torch.tensor(fix, dtype=_I32, device=device),
            "ohw": torch.tensor(ohw, dtype=_I32, device=device),
            "off": torch.tensor(offs, dtype=torch.int64, device=device),
            "sizes": ohw, "offsets": offs, "total": off, "maxpx": max(a * b for a, b in ohw)}


def rotate_expand_apply(x: torch.Tensor, plan, fill: int = 255, out: Optional[torch.Tensor] = None):
    n, h, w = _hwc(x, "rotate_expand.x")
    if plan["fix"].shape[0] != n:
        raise ValueError("rotate_expand: one angle per image")
    if out is None:
        out = torch.empty(plan["total"], dtype=_U8, device=x.device)
    elif out.numel() < plan["total"] or out.dtype != _U8:
        raise ValueError("rotate_expand: output buffer too small")
    _lib.call("lf_affine_nearest_fixed_u8", x.data_ptr(), out.data_ptr(), plan["fix"].data_ptr(),
              plan["ohw"].data_ptr(), plan["off"].data_ptr(), n, h, w, plan["maxpx"], int(fill),
              _stream())
    return out


def rotate_expand_u8(x: torch.Tensor, angles: Sequence[float], fill: int = 255):
    """Image.rotate(angle, expand=True, fillcolor=white) (NEAREST) for each image of a batch.

    Returns a list of [oh_i, ow_i, 3] uint8 views into one packed device buffer.
    """
    n, h, w = _hwc(x, "rotate_expand.x")
    if len(angles) != n:
        raise ValueError("rotate_expand: one angle per image")
    plan = rotate_expand_plan(w, h, angles, x.device)
    out = rotate_expand_apply(x, plan, fill)
    return [out[o:o + oh * ow * 3].view(oh, ow, 3) for o, (oh, ow) in zip(plan["offsets"], plan["sizes"])]


TILE_OUT, TILE_WINDOW, TILE_TAPS = 32, 48, 10   # lf_resample_tile_u8's tile, window and tap limits


def resample_tables_fit_tile(xb: np.ndarray, xk: np.ndarray, yb: np.ndarray, yk: np.ndarray, ow: int) -> bool:
    """Host check of lf_resample_tile_u8's preconditions on the (numpy) tables: at most 10 taps,
    ow % 4 == 0, and every run of 32 outputs reads at most 48 inputs on both axes."""
    if xk.shape[-1] > TILE_TAPS or yk.shape[-1] > TILE_TAPS or ow % 4:
        return False
    for b in (xb, yb):
        b = b.reshape(-1, b.shape[-2], 2).astype(np.int64)
        for o0 in range(0, b.shape[1], TILE_OUT):
            o1 = min(o0 + TILE_OUT, b.shape[1])
            span = (b[:, o0:o1, 0] + b[:, o0:o1, 1]).max(axis=1) - b[:, o0, 0]
            if int(span.max()) > TILE_WINDOW or (np.diff(b[:, o0:o1, 0], axis=1) < 0).any():
                return False
    return True


def resample_u8(x: torch.Tensor, oh: int, ow: int, xb: torch.Tensor, xk: torch.Tensor,
                yb: torch.Tensor, yk: torch.Tensor, per_image: bool, tile_ok: bool = False) -> torch.Tensor:
    """Pillow two-pass fixed-point resample with host-computed tables (int32 tensors).
    tile_ok: the caller checked `resample_tables_fit_tile` -> one fused kernel, no intermediate."""
    n, h, w = _hwc(x, "resample.x")
    for t, nm in ((xb, "xb"), (xk, "xk"), (yb, "yb"), (yk, "yk")):
        _chk(t, _I32, f"resample.{nm}")
    lead = (n,) if per_image else ()
    kx, ky = xk.shape[-1], yk.shape[-1]
    if (tuple(xb.shape) != lead + (ow, 2) or tuple(xk.shape) != lead + (ow, kx)
            or tuple(yb.shape) != lead + (oh, 2) or tuple(yk.shape) != lead + (oh, ky)):
        raise ValueError("resample: table shapes do not match (n, oh, ow)")
    out = torch.empty((n, oh, ow, 3), dtype=_U8, device=x.device)
    if tile_ok and kx <= TILE_TAPS and ky <= TILE_TAPS and ow % 4 == 0 and n <= 65535:
        _lib.call("lf_resample_tile_u8", x.data_ptr(), out.data_ptr(), n, h, w, oh, ow, xb.data_ptr(),
                  xk.data_ptr(), kx, yb.data_ptr(), yk.data_ptr(), ky, 1 if per_image else 0, _stream())
        return out
    tmp = torch.empty((n, h, ow, 3), dtype=_U8, device=x.device)
    _lib.call("lf_resample_u8", x.data_ptr(), tmp.data_ptr(), out.data_ptr(), n, h, w, oh, ow,
              xb.data_ptr(), xk.data_ptr(), kx, yb.data_ptr(), yk.data_ptr(), ky,
              1 if per_image else 0, _stream())
    return out


def resize_lanczos_u8(x: torch.Tensor, size: int) -> torch.Tensor:
    """ImageTransforms.resize_image(img, (size,size)) (LANCZOS) for a same-sized batch."""
    n, h, w = _hwc(x, "resize_lanczos.x")
    if (h, w) == (size, size):
        return x.clone()  # Image.resize returns a copy when nothing changes
    xb, xk, _ = _geo.lanczos_coeffs(w, 0.0, float(w), size)
    yb, yk, _ = _geo.lanczos_coeffs(h, 0.0, float(h), size)
    dev = x.device
    if w == size:  # Pillow skips the horizontal pass: identity table keeps the kernel generic
        xb = np.stack([np.arange(size), np.ones(size)], 1).astype(np.int32)
        xk = np.full((size, 1), 1 << _geo.PRECISION_BITS, dtype=np.int32)
    if h == size:
        yb = np.stack([np.arange(size), np.ones(size)], 1).astype(np.int32)
        yk = np.full((size, 1), 1 << _geo.PRECISION_BITS, dtype=np.int32)
    tile_ok = resample_tables_fit_tile(xb, xk, yb, yk, size)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (xb, xk, yb, yk)]
    return resample_u8(x, size, size, t[0], t[1], t[2], t[3], per_image=False, tile_ok=tile_ok)


def crop_resize_plan(w: int, h: int, boxes: Sequence[Sequence[int]], device):
    """Host-side Pillow resampling tables for per-image crops (device int32 tensors)."""
    tabs = [_geo.crop_resize_tables(w, h, *[int(v) for v in b]) for b in boxes]
    kx = max(t[2] for t in tabs)
    ky = max(t[5] for t in tabs)
    arrs = (np.stack([t[0] for t in tabs]), np.stack([_geo.pad_k(t[1], kx) for t in tabs]),
            np.stack([t[3] for t in tabs]), np.stack([_geo.pad_k(t[4], ky) for t in tabs]))
    tile_ok = resample_tables_fit_tile(arrs[0], arrs[1], arrs[2], arrs[3], w)
    return [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in arrs] + [tile_ok]


def crop_resize_lanczos_u8(x: torch.Tensor, boxes: Sequence[Sequence[int]]) -> torch.Tensor:
    """ImageAugmenter.crop: per image (left, top, nw, nh) crop then LANCZOS back to (W,H)."""
    n, h, w = _hwc(x, "crop_resize.x")
    if len(boxes) != n:
        raise ValueError("crop_resize: one box per image")
    t = crop_resize_plan(w, h, boxes, x.device)
    return resample_u8(x, h, w, t[0], t[1], t[2], t[3], per_image=True, tile_ok=t[4])
