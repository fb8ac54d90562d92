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


# ---------------------------------------------------------------------------------------------------------------------
# The Huffman step on the GPU: lf_jpeg_scan_prepare (host: markers, un-stuffing) + lf_jpeg_huffman_u8 (GPU)
# ---------------------------------------------------------------------------------------------------------------------
SCAN_KW = [dict(quality=95), dict(quality=60), dict(quality=95, optimize=True), dict(quality=85, restart_marker_rows=1),
           dict(quality=90, restart_marker_blocks=3), dict(quality=100)]


def _raw_scan(data):
    """The entropy-coded bytes of a one-scan file, from behind the SOS header up to (not including) EOI."""
    i = 2
    while data[i + 1] != 0xDA:
        i += 2 + int.from_bytes(data[i + 2:i + 4], "big")
    i += 2 + int.from_bytes(data[i + 2:i + 4], "big")
    assert data[-2:] == b"\xff\xd9"
    return data[i:-2]


def _prepared(data, h, w, extra=1 << 16):
    from leaffliction_amd.utils import jpeg_host
    slot = np.zeros((256 + 3 * h * w + extra + 15) // 16 * 16, np.uint8)
    got = jpeg_host.scan_prepare_into(data, slot)
    return slot, got


@pytest.mark.parametrize("kw", SCAN_KW)
def test_scan_prepare_keeps_every_bit_of_the_scan(kw):
    """What the host leaves for the GPU decoder, put back together (0xFF re-stuffed, RSTn re-inserted at the recorded
    offsets), is the file's own entropy-coded segment; header fields and tables are the file's."""
    from leaffliction_amd.utils import jpeg_host
    h, w = 64, 96
    a = make("noise", h, w, 31) if kw["quality"] == 100 else scene(h, w, 31)
    data = _save(a, **kw)
    slot, got = _prepared(data, h, w)
    assert got is not None and got[:2] == (h, w)
    aux = jpeg_host.scan_aux_offset(h, w)
    assert aux == (256 + 3 * h * w + 15) // 16 * 16
    hdr = slot[aux:aux + 32]
    assert bytes(hdr[:4]) == b"LFSC"
    hh, ww = hdr[4:8].view(np.uint16)
    restart, nint = (int(v) for v in hdr[8:16].view(np.uint32))
    data_off, data_len = (int(v) for v in hdr[24:32].view(np.uint32))
    assert (hh, ww) == (h, w) and int(hdr[16:24].view(np.uint64)[0]) == got[2]
    mcus = (h // 16) * (w // 16)
    assert nint == (-(-mcus // restart) if restart else 1)
    offs = slot[aux + 1120:aux + 1120 + 4 * (nint + 1)].view(np.uint32)
    assert offs[0] == 0 and offs[-1] == data_len and np.all(np.diff(offs.astype(np.int64)) > 0)
    body = slot[aux + data_off:aux + data_off + data_len]
    rebuilt = b""
    for i in range(nint):
        rebuilt += bytes(body[offs[i]:offs[i + 1]]).replace(b"\xff", b"\xff\x00")
        if i + 1 < nint:
            rebuilt += bytes([0xFF, 0xD0 + i % 8])
    assert rebuilt == _raw_scan(data)
    # the tables the reader would have used
    ref = jpeg_host.read_file(data)
    assert np.array_equal(slot[:256].view(np.uint16).reshape(2, 64), ref[1])
    # same tables -> same hash; other tables -> another hash
    if kw.get("optimize"):
        assert _prepared(_save(a, quality=95), h, w)[1][2] != got[2]
    else:
        assert _prepared(_save(scene(h, w, 32), **kw), h, w)[1][2] == got[2]


def test_scan_prepare_declines_what_the_host_reader_declines():
    a = scene(64, 64, 12)
    for data in (_save(a, quality=90, subsampling=0), _save(a, quality=90, progressive=True), _save(a[..., 0], quality=90),
                 _save(a[:50, :60], quality=90), b"not a jpeg at all", _save(a, quality=90)[:400]):
        assert _prepared(data, 64, 64)[1] is None
    good = _save(a, quality=90)
    assert _prepared(good, 64, 64)[1] is not None
    assert _prepared(good[:-2], 64, 64)[1] is None            # no EOI behind the scan
    assert _prepared(good[:len(good) * 3 // 4], 64, 64)[1] is None
    assert _prepared(good + b"\x00" * 7, 64, 64)[1] is not None
    assert _prepared(good, 64, 64, extra=256)[1] is None      # does not fit the slot
    rst = _save(a, quality=90, restart_marker_rows=1)
    k = rst.index(b"\xff\xd1")
    assert _prepared(rst[:k] + rst[k + 2:], 64, 64)[1] is None   # a restart marker went missing


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,kw", [(224, 224, dict(quality=95)), (64, 96, dict(quality=70, optimize=True)),
                                    (48, 208, dict(quality=85, restart_marker_rows=1)), (16, 16, dict(quality=95)),
                                    (64, 64, dict(quality=90, restart_marker_blocks=3)), (32, 48, dict(quality=100)),
                                    (256, 256, dict(quality=88)), (224, 224, dict(quality=30)),
                                    (64, 64, dict(quality=100, optimize=True)),
                                    (320, 336, dict(quality=100))])   # noise at quality 100: scans over the 96 KB LDS stage
@pytest.mark.parametrize("sequential", [False, True])
def test_gpu_huffman_decoder_recovers_the_coefficients(cuda, h, w, kw, sequential):
    """file -> host markers -> GPU Huffman decoding == the host reader's coefficients, and on through the GPU IDCT ==
    Image.open(file).convert("RGB"), through the workgroup-per-image kernel (256 subsequences decoded at once; files
    with restart markers fall through to the other kernel) and through the lane-per-image kernel alone.  70 images:
    two groups of 64 lanes, the second one ragged."""
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    n = 70
    kinds = ["noise", "scene", "extremes", "flat"]
    files = [_save(make(kinds[i % 4], h, w, 40 + i), **kw) for i in range(n)]
    if kw.get("optimize") and sequential:   # optimised tables differ from file to file, and that kernel shares them
        files = [files[1]] * n
    stride = (256 + 3 * h * w + max(1 << 16, max(len(f) for f in files) + 4096) + 4095) // 4096 * 4096
    slots = np.zeros((n, stride), np.uint8)
    for i, f in enumerate(files):
        got = jpeg_host.scan_prepare_into(f, slots[i])
        assert got is not None and got[:2] == (h, w), i
    if (h, w) == (320, 336):
        assert max(len(f) for f in files) > 100 * 1024
    dev = torch.from_numpy(slots).to(cuda)
    status = ops.jpeg_huffman_u8(dev, h, w, sequential=sequential).cpu().numpy()
    assert np.all(status == 0), status
    out = dev.cpu().numpy()
    m = (h // 16) * (w // 16)
    for i, f in enumerate(files):
        coef = jpeg_host.read_file(f)[0]
        assert np.array_equal(out[i, 256:256 + m * 768].view(np.int16).reshape(m, 6, 64), coef), i
    rgb = ops.jpeg_idct_rgb_u8(dev, h, w).cpu().numpy()
    for i in (0, 1, 2, 3, 63, 64, 69):
        assert np.array_equal(rgb[i], np.asarray(Image.open(io.BytesIO(files[i])).convert("RGB"))), i


@pytest.mark.gpu
@pytest.mark.parametrize("sequential", [False, True])
def test_gpu_huffman_decoder_reports_what_it_cannot_decode(cuda, sequential):
    """status 1: a scan that ends early (bytes cut out of its middle with the EOI left in place — what a damaged
    file looks like after the host's marker pass), a restart interval that is too short; status 2 (lane-per-image
    kernel): tables other than the group's; status 3: a slot nothing was prepared in.  The neighbours decode all
    the same."""
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    h = w = 64
    stride = (256 + 3 * h * w + (1 << 15) + 4095) // 4096 * 4096
    good = _save(scene(h, w, 50), quality=95)
    cut = good[:len(good) // 2] + good[-2:]                       # half the scan gone, EOI in place
    other = _save(scene(h, w, 51), quality=95, optimize=True)     # its own Huffman tables
    rst = _save(scene(h, w, 52), quality=95, restart_marker_rows=1)
    k = rst.index(b"\xff\xd1")
    rst_short = rst[:k - 30] + rst[k:]                            # 30 bytes gone from the second interval
    cut2 = good[:len(good) - 40] + good[-2:]                      # only the last few blocks gone
    junk = good[:len(good) // 3] + bytes(255 - b if b not in (0, 255) else b for b in good[len(good) // 3:-2]) + good[-2:]
    files = [good, cut, other, good, None, rst, rst_short, good, cut2, junk]
    slots = np.zeros((len(files), stride), np.uint8)
    for i, f in enumerate(files):
        if f is not None:
            assert jpeg_host.scan_prepare_into(f, slots[i]) is not None, i
    dev = torch.from_numpy(slots).to(cuda)
    status = ops.jpeg_huffman_u8(dev, h, w, sequential=sequential).cpu().numpy()
    # rst / rst_short have the standard tables too (same hash as `good`), so they decode in this group; `junk` (the
    # bits of two thirds of the scan inverted) is whatever the host reader says it is
    junk_ok = jpeg_host.read_file(junk) is not None
    assert status.tolist() == [0, 1, 2 if sequential else 0, 0, 3, 0, 1, 0, 1, 0 if junk_ok else 1], status
    out = dev.cpu().numpy()
    m = (h // 16) * (w // 16)
    for i in (0, 3, 5, 7) + (() if sequential else (2,)) + ((9,) if junk_ok else ()):
        assert np.array_equal(out[i, 256:256 + m * 768].view(np.int16).reshape(m, 6, 64), jpeg_host.read_file(files[i])[0])
    for f in (cut, rst_short, cut2):   # ... the host reader declines them too: such files go to libjpeg, whose verdict
        assert jpeg_host.read_file(f) is None   # (error, or pixels with its own concealment) is the reference's
    # a group whose FIRST slot holds nothing: no tables to decode with
    dev2 = torch.from_numpy(slots[[4, 0]]).to(cuda)
    assert ops.jpeg_huffman_u8(dev2, h, w, sequential=sequential).cpu().numpy().tolist() == [3, 3 if sequential else 0]


@pytest.mark.gpu
def test_gpu_encoder_takes_images_of_different_sizes_in_one_launch(cuda):
    """ops.jpeg_encode_items_u8: the balancer's rotated canvases — every one its own size, whole MCUs or not — lie in
    slots of one buffer and leave as finished scans in the same places; with the markers around them the files are
    Pillow's, byte for byte."""
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    sizes = [(224, 224), (241, 263), (17, 9), (16, 16), (300, 187), (1, 1), (64, 95), (256, 256), (33, 48), (224, 224)]
    room = (2 * 224 * 224 * 3 + 4095) // 4096 * 4096
    imgs = [make(["scene", "noise", "extremes", "flat"][i % 4], h, w, 70 + i) for i, (h, w) in enumerate(sizes)]
    host = np.zeros((len(sizes), room), np.uint8)
    for i, a in enumerate(imgs):
        host[i, :a.size] = a.reshape(-1)
    dev = torch.from_numpy(host).to(cuda)
    ops.jpeg_encode_items_u8(dev.view(-1), [(i * room, h, w) for i, (h, w) in enumerate(sizes)], room, 95)
    out = dev.cpu().numpy()
    for i, ((h, w), a) in enumerate(zip(sizes, imgs)):
        n = int(out[i, :4].view(np.int32)[0])
        assert n > 0, (i, n)
        assert jpeg_host.wrap_scan(out[i, 4:4 + n], h, w, 95) == pil_bytes(a, 95), (i, h, w)
    # a place that is too small for its scan reports it
    small = torch.from_numpy(host[1:2].copy()).to(cuda)
    ops.jpeg_encode_items_u8(small.view(-1), [(0, 241, 263)], 2048, 95)
    assert int(small.cpu().numpy()[0, :4].view(np.int32)[0]) == -1
    with pytest.raises(ValueError):
        ops.jpeg_encode_items_u8(dev.view(-1), [(2, 16, 16)], room, 95)
