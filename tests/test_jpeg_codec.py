"""JPEG encode: the oracle (oracle/jpeg_ref.py, libjpeg's integer pipeline restated) is pinned to the BYTES
Pillow writes — Pillow is the reference's own encoder (srcs/utils/image_utils.py:49-56) and is installed —,
the host entropy coder (libleafcodec.so) to the oracle and to Pillow, and (GPU) the coefficients of
lf_jpeg_fdct_quant_u8 to the oracle's and the finished files to Pillow's, byte for byte."""
import io

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_ref as J


def pil_bytes(a, quality=95):
    b = io.BytesIO()
    Image.fromarray(a).save(b, format="JPEG", quality=quality)
    return b.getvalue()


def scene(h, w, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 17.0 + seed) * np.cos(yy / 23.0), 90 + 80 * np.cos(xx / 9.0),
                    140 + 60 * np.sin((xx + yy) / 31.0)], -1) + r.normal(0, 3 + 4 * (seed % 3), (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def mcu_order(y, cb, cr):
    """oracle planes of blocks -> [MCUs, 6, 64] in scan order"""
    my, mx = cb.shape[:2]
    out = np.zeros((my * mx, 6, 64), np.int16)
    for i in range(my):
        for j in range(mx):
            m = out[i * mx + j]
            m[0], m[1], m[2], m[3] = y[2 * i, 2 * j], y[2 * i, 2 * j + 1], y[2 * i + 1, 2 * j], y[2 * i + 1, 2 * j + 1]
            m[4], m[5] = cb[i, j], cr[i, j]
    return out


CASES = [("noise", 32, 48, 0), ("scene", 64, 64, 1), ("scene", 224, 224, 2), ("flat", 16, 16, 3), ("extremes", 48, 32, 4)]


def make(kind, h, w, seed):
    rng = np.random.RandomState(seed)
    if kind == "noise":
        return rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if kind == "flat":
        return np.full((h, w, 3), 200, np.uint8)
    if kind == "extremes":   # saturated checkerboards: the largest coefficients, long zero runs, 0xFF bytes to stuff
        yy, xx = np.mgrid[0:h, 0:w]
        a = (((yy // 3 + xx // 5) % 2) * 255).astype(np.uint8)
        return np.stack([a, 255 - a, np.where(xx < w // 2, a, 0).astype(np.uint8)], -1)
    return scene(h, w, seed)


@pytest.mark.parametrize("kind,h,w,seed", CASES)
def test_oracle_writes_pillows_bytes(kind, h, w, seed):
    a = make(kind, h, w, seed)
    assert J.encode(a) == pil_bytes(a)


@pytest.mark.parametrize("quality", [50, 75, 90, 100])
def test_oracle_other_qualities(quality):
    a = scene(48, 64, quality)
    assert J.encode(a, quality) == pil_bytes(a, quality)


@pytest.mark.parametrize("kind,h,w,seed", CASES)
def test_host_entropy_coder_writes_pillows_bytes(kind, h, w, seed):
    from leaffliction_amd.utils import jpeg_host
    a = make(kind, h, w, seed)
    coef = mcu_order(*J.quantised_coefficients(a))
    assert jpeg_host.write_file(coef, h, w) == pil_bytes(a)


def test_host_coder_rejects_bad_shapes():
    from leaffliction_amd.utils import jpeg_host
    with pytest.raises(ValueError):
        jpeg_host.write_file(np.zeros((1, 6, 64), np.int16), 16, 32)
    with pytest.raises(ValueError):
        jpeg_host.write_file(np.zeros((1, 6, 64), np.int16), 17, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,h,w,seed", CASES + [("scene", 96, 208, 7), ("noise", 224, 224, 8)])
def test_gpu_coefficients_and_files(cuda, kind, h, w, seed):
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    batch = np.stack([make(kind, h, w, seed + 10 * i) for i in range(3)])
    coef = ops.jpeg_fdct_quant_u8(torch.from_numpy(batch).to(cuda)).cpu().numpy()
    for i in range(3):
        assert np.array_equal(coef[i], mcu_order(*J.quantised_coefficients(batch[i]))), i
        assert jpeg_host.write_file(coef[i], h, w) == pil_bytes(batch[i]), i


@pytest.mark.gpu
def test_gpu_other_quality_and_bad_shape(cuda):
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    a = scene(64, 80, 5)
    coef = ops.jpeg_fdct_quant_u8(torch.from_numpy(a[None]).to(cuda), quality=80).cpu().numpy()
    assert jpeg_host.write_file(coef[0], 64, 80, 80) == pil_bytes(a, 80)
    with pytest.raises(ValueError):
        ops.jpeg_idct_rgb_u8(torch.zeros((1, 4096), dtype=torch.uint8, device=cuda), 30, 32)   # decoding: whole MCUs only


# ---- reading -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,h,w,seed", CASES)
def test_oracle_decoder_stages_give_pillows_pixels(kind, h, w, seed):
    a = make(kind, h, w, seed)
    want = np.asarray(Image.open(io.BytesIO(pil_bytes(a))).convert("RGB"))
    assert np.array_equal(J.decode_coefficients(*J.quantised_coefficients(a)), want)


def _save(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).save(b, format="JPEG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("kw", [dict(quality=95), dict(quality=60), dict(quality=95, optimize=True),
                                dict(quality=85, restart_marker_rows=1), dict(quality=90, restart_marker_blocks=3)])
def test_host_reader_recovers_the_coefficients(kw):
    """Files written by Pillow itself (standard and optimised Huffman tables, restart markers): the reader's
    coefficients and tables are the encoder's."""
    from leaffliction_amd.utils import jpeg_host
    a = scene(64, 96, 11)
    got = jpeg_host.read_file(_save(a, **kw))
    assert got is not None
    coef, qtab, h, w = got
    q = kw["quality"]
    assert (h, w) == (64, 96)
    assert np.array_equal(coef, mcu_order(*J.quantised_coefficients(a, q)))
    ql, qc = J.quant_tables(q)
    assert np.array_equal(qtab[0], ql) and np.array_equal(qtab[1], qc)


def test_host_reader_declines_what_it_does_not_cover():
    from leaffliction_amd.utils import jpeg_host
    a = scene(64, 64, 12)
    assert jpeg_host.read_file(_save(a, quality=90, subsampling=0)) is None          # 4:4:4
    assert jpeg_host.read_file(_save(a, quality=90, progressive=True)) is None
    assert jpeg_host.read_file(_save(a[..., 0], quality=90)) is None                   # greyscale
    assert jpeg_host.read_file(_save(a[:50, :60], quality=90)) is None                # not whole MCUs
    assert jpeg_host.read_file(b"not a jpeg at all") is None
    assert jpeg_host.read_file(_save(a, quality=90)[:400]) is None                     # truncated


@pytest.mark.parametrize("kw", [dict(quality=90), dict(quality=95, restart_marker_rows=1)])
def test_host_reader_hands_back_files_cut_inside_the_scan(kw):
    """A file truncated INSIDE its entropy-coded scan (or missing its EOI) must not come back as pixels: Pillow raises
    "image file is truncated" for it, which is what makes the reference count the task as failed / skip the file
    (image_utils.py:19-33).  The reader used to decode the missing MCUs from padding zeros and report success; it
    now declines, and the caller's libjpeg path gives the reference's verdict."""
    import io
    from PIL import Image
    from leaffliction_amd.utils import jpeg_host
    a = scene(224, 224, 5)
    data = _save(a, **kw)
    assert jpeg_host.read_file(data) is not None                                       # the intact file is taken
    for cut in (len(data) // 2, len(data) * 3 // 4, len(data) - 40, len(data) - 3, len(data) - 2, len(data) - 1):
        assert jpeg_host.read_file(data[:cut]) is None, cut
        with pytest.raises(OSError):                                                   # ... and Pillow refuses it
            Image.open(io.BytesIO(data[:cut])).convert("RGB")
    # a marker planted in the middle of the scan ends it early: declined as well
    mid = len(data) // 2
    assert jpeg_host.read_file(data[:mid] + b"\xff\xd9" + data[mid + 2:]) is None
    # trailing bytes behind EOI are harmless (Pillow reads such files)
    assert jpeg_host.read_file(data + b"\x00" * 7) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,kw", [(224, 224, dict(quality=95)), (64, 96, dict(quality=70, optimize=True)),
                                    (48, 208, dict(quality=85, restart_marker_rows=1)), (16, 16, dict(quality=95)),
                                    (256, 256, dict(quality=88))])
def test_gpu_decode_gives_pillows_pixels(cuda, h, w, kw):
    """file -> host Huffman decoding -> GPU IDCT / upsampling / colour == Image.open(file).convert("RGB")."""
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    files = [_save(make("scene" if i else "noise", h, w, 20 + i), **kw) for i in range(3)]
    stride = (256 + 3 * h * w + 4095) // 4096 * 4096
    slots = np.zeros((3, stride), np.uint8)
    for i, f in enumerate(files):
        assert jpeg_host.read_file_into(f, slots[i]) == (h, w)
    got = ops.jpeg_idct_rgb_u8(torch.from_numpy(slots).to(cuda), h, w).cpu().numpy()
    for i, f in enumerate(files):
        assert np.array_equal(got[i], np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))), i


@pytest.mark.gpu
@pytest.mark.parametrize("kind,h,w,seed", CASES + [("scene", 96, 208, 7), ("noise", 224, 224, 8), ("scene", 256, 256, 9)])
def test_gpu_entropy_coder_writes_pillows_bytes(cuda, kind, h, w, seed):
    """pixels -> GPU colour / DCT / quantisation -> GPU Huffman coding and byte stuffing -> markers == Pillow's file."""
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    batch = np.stack([make(kind, h, w, seed + 10 * i) for i in range(3)])
    coef = ops.jpeg_fdct_quant_u8(torch.from_numpy(batch).to(cuda))
    rows = ops.jpeg_entropy_u8(coef, h, w).cpu().numpy()
    for i in range(3):
        n = int(rows[i, :4].view(np.int32)[0])
        assert n > 0
        assert jpeg_host.wrap_scan(rows[i, 4:4 + n], h, w) == pil_bytes(batch[i]), i


@pytest.mark.gpu
def test_gpu_entropy_coder_reports_a_row_that_is_too_small(cuda):
    import torch
    from leaffliction_amd import ops
    a = make("noise", 64, 64, 3)
    coef = ops.jpeg_fdct_quant_u8(torch.from_numpy(a[None]).to(cuda))
    rows = ops.jpeg_entropy_u8(coef, 64, 64, out_stride=1024).cpu().numpy()
    assert int(rows[0, :4].view(np.int32)[0]) == -1


# ---- ragged sizes (rotated outputs): libjpeg's padding ----------------------------------------------------
RAGGED = [(17, 23), (30, 50), (100, 75), (225, 225), (224, 230), (8, 8), (1, 1), (16, 33), (31, 16), (47, 47), (291, 283)]


def mcu_order_ragged(y, cb, cr):
    return mcu_order(y, cb, cr)   # the oracle already returns 2*My x 2*Mx luminance blocks


@pytest.mark.parametrize("h,w", RAGGED)
def test_oracle_and_host_coder_on_ragged_sizes(h, w):
    from leaffliction_amd.utils import jpeg_host
    for a in (scene(h, w, h + w), np.random.RandomState(h * w).randint(0, 256, (h, w, 3)).astype(np.uint8)):
        want = pil_bytes(a)
        assert J.encode(a) == want
        assert jpeg_host.write_file(mcu_order_ragged(*J.quantised_coefficients(a)), h, w) == want


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", RAGGED)
def test_gpu_encoder_on_ragged_sizes(cuda, h, w):
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    batch = np.stack([scene(h, w, h + w + i) if i else np.random.RandomState(h + w).randint(0, 256, (h, w, 3)).astype(np.uint8)
                      for i in range(3)])
    coef = ops.jpeg_fdct_quant_u8(torch.from_numpy(batch).to(cuda))
    rows = ops.jpeg_entropy_u8(coef, h, w).cpu().numpy()
    coef = coef.cpu().numpy()
    for i in range(3):
        assert np.array_equal(coef[i], mcu_order_ragged(*J.quantised_coefficients(batch[i]))), i
        n = int(rows[i, :4].view(np.int32)[0])
        assert n > 0 and jpeg_host.wrap_scan(rows[i, 4:4 + n], h, w) == pil_bytes(batch[i]), i


# ---- the distortion op's noise plane in C -----------------------------------------------------------------
@pytest.mark.parametrize("seed", [1, 7, 12345, 999999, 2 ** 31 + 5, 0])
def test_legacy_normal_equals_numpy(seed):
    """np.random.RandomState(seed).normal(0, 5, n): the float64 stream bit for bit, and its uint8 cast."""
    from leaffliction_amd.utils import jpeg_host
    n = 224 * 224 * 3 if seed != 7 else 1001     # odd counts leave half a pair unused
    want = np.random.RandomState(seed).normal(0, 5, n)
    got = jpeg_host.legacy_normal_f64(seed, 0.0, 5.0, n)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    out = np.empty(n, np.uint8)
    jpeg_host.legacy_normal_u8(seed, 0.0, 5.0, out)
    assert np.array_equal(out, want.astype(np.uint8))
