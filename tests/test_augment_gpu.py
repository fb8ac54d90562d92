"""HIP augmentation kernels (through the C ABI) vs the oracle and the golden vectors.

Bit-exact for every uint8 / int32 result; the f32 pack is bit-exact vs numpy f32.
"""
import random

import numpy as np
import pytest

from conftest import leaf_like
from oracle import cv_ops as CV
from oracle import pil_ops as P

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def batch_inputs(n, h, w, seed):
    rng = np.random.RandomState(seed)
    imgs = [leaf_like(h, w, seed + i) if i % 2 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
            for i in range(n)]
    return np.stack(imgs)


SIZES = [(224, 224), (64, 48), (33, 17), (5, 7)]  # aligned fast path, ragged fallbacks, tiny


@pytest.mark.parametrize("h,w", SIZES)
def test_pack_bit_exact(cuda, h, w):
    from leaffliction_amd import ops
    x = batch_inputs(3, h, w, 1)
    got = ops.pack_hwc_u8_to_nchw_f32(dev(x, cuda)).cpu().numpy()
    assert np.array_equal(got, P.pack_nchw(x))
    mean, den = [0.4, 0.5, 0.3], [0.2, 0.25, 0.22]
    got = ops.pack_hwc_u8_to_nchw_f32(dev(x, cuda), mean, den).cpu().numpy()
    assert np.array_equal(got, P.pack_nchw(x, mean, den))


@pytest.mark.parametrize("h,w", SIZES)
def test_hist_lut_autocontrast_bit_exact(cuda, h, w):
    import torch
    from leaffliction_amd import ops
    x = batch_inputs(5, h, w, 2)
    x[4] = 91  # flat image: hi <= lo -> identity LUT
    xd = dev(x, cuda)
    hist = ops.hist_u8(xd).cpu().numpy()
    for i in range(5):
        assert np.array_equal(hist[i], P.histogram(x[i]))
    cut = np.array([0.0, 0.37, 1.99, 1.0, 0.5])
    lut = ops.autocontrast_lut(ops.hist_u8(xd), dev(cut, cuda)).cpu().numpy()
    out = ops.autocontrast_u8(xd, dev(cut, cuda)).cpu().numpy()
    for i in range(5):
        assert np.array_equal(lut[i], P.autocontrast_lut(P.histogram(x[i]), float(cut[i])))
        assert np.array_equal(out[i], P.autocontrast(x[i], float(cut[i])))


@pytest.mark.parametrize("h,w", SIZES)
def test_flip_noise_mask_bit_exact(cuda, h, w):
    from leaffliction_amd import ops
    x = batch_inputs(4, h, w, 3)
    xd = dev(x, cuda)
    mode = np.array([0, 1, 1, 0], np.int32)
    got = ops.flip_u8(xd, dev(mode, cuda)).cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], P.flip(x[i], int(mode[i])))
    noise = np.random.RandomState(9).normal(0, 5, x.shape)
    noise[0, 0, 0] = [-256.2, 300.7, -3.7]
    assert np.array_equal(ops.noise_wrap_add_u8(xd, dev(noise, cuda)).cpu().numpy(),
                          P.noise_wrap_add(x, noise))
    mask = np.random.RandomState(4).randint(0, 256, (4, h, w)).astype(np.uint8)
    mask[0] = 127
    mask[1] = 128
    for color in ("white", "black"):
        got = ops.mask_composite_u8(xd, dev(mask, cuda), color).cpu().numpy()
        for i in range(4):
            assert np.array_equal(got[i], CV.apply_mask(x[i], mask[i], color))
    with pytest.raises(ValueError):
        ops.mask_composite_u8(xd, dev(mask, cuda), "red")


@pytest.mark.parametrize("h,w", [(224, 224), (64, 48), (33, 17), (16, 16)])
def test_distortion_fused_pass_equals_the_separate_kernels(cuda, h, w):
    """ImageAugmenter.distortion (image_augmenter.py:121-131): the kernel that adds the noise AND counts the noisy
    image in the same pass gives the bytes of the add kernel and the bins of the histogram kernel — for the host's
    uint8 noise plane (bit-exact against the oracle's autocontrast of the wrapped sum) and for the device-drawn
    Philox noise (same seed, same bytes); 33 x 17 is not a multiple of 16 bytes and takes the separate kernels."""
    from leaffliction_amd import ops
    n = 4   # 4 x 33 x 17 x 3 bytes is still a multiple of 4: the separate add kernel takes it
    x = batch_inputs(n, h, w, 13)
    xd = dev(x, cuda)
    n8 = np.random.RandomState(2).normal(0, 5, x.shape).astype(np.uint8)   # numpy's own cast, as the reference does
    cut = np.array([0.0, 0.4, 1.3, 1.99])
    y, hist = ops.noise_hist_u8(xd, dev(n8, cuda))
    assert np.array_equal(y.cpu().numpy(), (x + n8).astype(np.uint8))
    assert np.array_equal(hist.cpu().numpy(), ops.hist_u8(y).cpu().numpy())
    got = ops.distortion_u8(xd, dev(cut, cuda), add=dev(n8, cuda)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], P.autocontrast((x[i] + n8[i]).astype(np.uint8), float(cut[i]))), i
    yp, hp = ops.noise_hist_u8(xd, None, seed=42, sigma=5.0)
    ref = ops.noise_philox_add_u8(xd, 42, 5.0)
    assert np.array_equal(yp.cpu().numpy(), ref.cpu().numpy())
    assert np.array_equal(hp.cpu().numpy(), ops.hist_u8(ref).cpu().numpy())


def test_apply_mask_docstring_example(cuda):
    """srcs/utils/mask_utils.py:29-41."""
    from leaffliction_amd import ops
    img = np.random.randint(0, 255, (1, 100, 100, 3), dtype=np.uint8)
    mask = np.random.randint(0, 2, (1, 100, 100), dtype=np.uint8) * 255
    w = ops.mask_composite_u8(dev(img, cuda), dev(mask, cuda), "white").cpu().numpy()
    b = ops.mask_composite_u8(dev(img, cuda), dev(mask, cuda), "black").cpu().numpy()
    assert (w[0][mask[0] == 0] == 255).all() and (b[0][mask[0] == 0] == 0).all()
    assert np.array_equal(w[0][mask[0] == 255], img[0][mask[0] == 255])


def test_philox_noise_statistics(cuda):
    """Device-drawn noise: integer noise must follow trunc(N(0,5)) statistically."""
    from leaffliction_amd import ops
    x = np.full((16, 224, 224, 3), 128, np.uint8)
    got = ops.noise_philox_add_u8(dev(x, cuda), seed=42, sigma=5.0).cpu().numpy().astype(np.int64)
    d = got - 128
    assert abs(d.mean()) < 0.05
    assert abs(d.std() - np.trunc(np.random.RandomState(0).normal(0, 5, 2_000_000)).std()) < 0.05
    again = ops.noise_philox_add_u8(dev(x, cuda), seed=42, sigma=5.0).cpu().numpy()
    assert np.array_equal(again, got.astype(np.uint8))  # counter-based: reproducible


@pytest.mark.parametrize("h,w", SIZES)
def test_colour_conversions_bit_exact(cuda, h, w):
    from leaffliction_amd import ops
    x = batch_inputs(3, h, w, 5)
    xd = dev(x, cuda)
    assert np.array_equal(ops.rgb2hsv_u8(xd).cpu().numpy(), CV.rgb2hsv(x))
    assert np.array_equal(ops.rgb2gray_u8(xd).cpu().numpy(), CV.rgb2gray(x))
    counts, hh = ops.hsv_region_stats(xd)
    counts, hh = counts.cpu().numpy(), hh.cpu().numpy()
    for i in range(3):
        c, hist = CV.hsv_region_stats(x[i])
        assert np.array_equal(counts[i], c)
        assert np.array_equal(hh[i], hist)


def test_hsv_all_colours(cuda):
    """Every 8-bit RGB triple on a 32-step lattice plus the cube corners and greys."""
    from leaffliction_amd import ops
    v = np.arange(0, 256, 5, dtype=np.uint8)
    grid = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(1, -1, 1, 3)
    grid = np.ascontiguousarray(np.concatenate([grid, grid[:, :4 - grid.shape[1] % 4]], 1))
    assert np.array_equal(ops.rgb2hsv_u8(dev(grid, cuda)).cpu().numpy(), CV.rgb2hsv(grid))


@pytest.mark.parametrize("h,w", [(224, 224), (64, 48), (33, 17), (256, 256), (32, 32), (64, 352), (96, 64)])
@pytest.mark.parametrize("ksize,sigma", [(15, 0.0), (5, 1.5), (3, 0.0), (5, 0.0), (7, 0.0), (11, 2.0), (13, 0.0),
                                         (9, 0.6)])
def test_gauss_blur_bit_exact(cuda, h, w, ksize, sigma):
    """cv2.GaussianBlur's fixed-point arithmetic, every pixel.  Heights and row lengths that are multiples of
    32 with taps below 128 go to the i8-MFMA kernel (2 k-steps up to 3*R <= 16, else 3; one, several or a
    prime number of 32-byte columns per row; a single 32-row block, where both reflections act on the same
    block); everything else (ragged shapes, the 128 tap of the 3-tap kernel, the 0.6-sigma kernel's centre tap)
    to the dot-product kernels."""
    from leaffliction_amd import ops
    x = batch_inputs(2, h, w, 6)
    got = ops.gauss_blur_u8(dev(x, cuda), ksize, sigma).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], CV.gaussian_blur(x[i], ksize, sigma))
    g = np.ascontiguousarray(x[..., 1])
    got = ops.gauss_blur_u8(dev(g, cuda), ksize, sigma).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], CV.gaussian_blur(g[i], ksize, sigma))


@pytest.mark.parametrize("h,w", [(224, 224), (64, 48), (33, 17), (416, 400)])
def test_blur_saliency_bit_exact(cuda, h, w):
    """apply_blur_filter (blur.py:18-79) given the leaf mask: every pixel equal to the oracle's.
    416x400 does not fit the LDS copy of the Canny map (hysteresis sweeps in global memory)."""
    from leaffliction_amd import ops
    n = 2 if h > 300 else 4
    x = batch_inputs(n, h, w, 9)
    yy, xx = np.mgrid[0:h, 0:w]
    rng = np.random.RandomState(3)
    masks = np.stack([(((yy - h // 2) ** 2 + (xx - w // 2) ** 2 <= (min(h, w) * (0.3 + 0.05 * i)) ** 2)
                       * rng.choice([1, 200, 255])).astype(np.uint8) for i in range(n)])
    got = ops.blur_saliency_u8(dev(x, cuda), dev(masks, cuda)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], CV.blur_saliency(x[i], masks[i])), i
    got = ops.blur_saliency_u8(dev(x, cuda), dev(masks, cuda), gaussian_sigma=0.8, brown_hue_range=(5, 25),
                               brown_s_min=40, brown_v_max=180).cpu().numpy()
    assert np.array_equal(got[1], CV.blur_saliency(x[1], masks[1], 0.8, (5, 25), 40, 180))
    got = ops.blur_saliency_u8(dev(x, cuda), dev(masks, cuda), use_brown=False).cpu().numpy()
    assert np.array_equal(got[0], CV.blur_saliency(x[0], masks[0], use_brown=False))


def test_golden_augmenter_cases(cuda, golden):
    """Every reference ImageAugmenter output (42 cases) reproduced bit-exactly on the GPU."""
    import torch
    from leaffliction_amd.preprocessing import image_augmenter as IA
    arrays, meta = golden
    for c in meta["cases"]:
        img = arrays[c["input"]]
        h, w, _ = img.shape
        random.seed(c["seed"])
        np.random.seed(c["seed"])
        p = IA.draw_params(c["op"], w, h)
        out = IA.apply_batch(c["op"], dev(img[None], cuda), [p])[0].cpu().numpy()
        exp = arrays[c["output"]]
        assert out.shape == exp.shape, c
        assert np.array_equal(out, exp), c


def test_golden_loader_resize(cuda, golden):
    from leaffliction_amd import ops
    arrays, _ = golden
    for si in range(3):
        img = arrays[f"in_{si}"]
        for S in (32, 64, 224):
            got = ops.resize_lanczos_u8(dev(img[None], cuda), S)[0].cpu().numpy()
            assert np.array_equal(got, arrays[f"resize_{si}_{S}"]), (si, S)


def test_geometric_batch_vs_oracle(cuda):
    """Batched, per-image parameters; also size-independent properties at 224x224."""
    import torch
    from leaffliction_amd import ops
    from leaffliction_amd.preprocessing import image_augmenter as IA
    x = batch_inputs(6, 224, 224, 11)
    xd = dev(x, cuda)
    random.seed(5)
    np.random.seed(5)
    for op in ("rotate", "skew", "shear", "crop"):
        ps = [IA.draw_params(op, 224, 224) for _ in range(6)]
        outs = IA.apply_batch(op, xd, ps)
        for i in range(6):
            if op == "rotate":
                exp = P.rotate_expand_white(x[i], ps[i]["angle"])
            elif op == "crop":
                exp = P.crop_resize_lanczos(x[i], *ps[i]["box"])
            else:
                exp = P.warp_bicubic(x[i], ps[i]["coeffs"], op == "skew")
            assert np.array_equal(outs[i].cpu().numpy(), exp), (op, i)
    # the dataset's native 256x256 and a non-square size (rows != columns in every index formula)
    for (hh, ww, seed) in ((256, 256, 21), (150, 260, 22)):
        xs = batch_inputs(3, hh, ww, seed)
        xsd = dev(xs, cuda)
        random.seed(seed)
        np.random.seed(seed)
        for op in ("flip", "rotate", "skew", "shear", "crop", "distortion"):
            ps = [IA.draw_params(op, ww, hh) for _ in range(3)]
            outs = IA.apply_batch(op, xsd, ps)
            for i in range(3):
                if op == "flip":
                    exp = P.flip(xs[i], ps[i]["mode"])
                elif op == "rotate":
                    exp = P.rotate_expand_white(xs[i], ps[i]["angle"])
                elif op == "crop":
                    exp = P.crop_resize_lanczos(xs[i], *ps[i]["box"])
                elif op == "distortion":
                    exp = P.autocontrast(P.noise_wrap_add(xs[i], ps[i]["noise"]), ps[i]["cutoff"])
                else:
                    exp = P.warp_bicubic(xs[i], ps[i]["coeffs"], op == "skew")
                assert np.array_equal(outs[i].cpu().numpy(), exp), (op, i, hh, ww)
    # flip is an involution; identity warp reproduces the input
    mode = torch.zeros(6, dtype=torch.int32, device=cuda)
    assert torch.equal(ops.flip_u8(ops.flip_u8(xd, mode), mode), xd)
    ident = torch.tensor([[1, 0, 0, 0, 1, 0, 0, 0]] * 6, dtype=torch.float64, device=cuda)
    assert torch.equal(ops.warp_bicubic_u8(xd, ident, False), xd)
    # histogram totals: every bin count sums to H*W per channel
    assert (ops.hist_u8(xd).sum(-1) == 224 * 224).all()


@pytest.mark.parametrize("h,w", [(224, 224), (256, 256), (150, 260), (64, 48), (40, 36), (33, 20)])
def test_fused_resample_matches_two_pass_and_oracle(cuda, h, w):
    """lf_resample_tile_u8 (both LANCZOS passes in one kernel) == lf_resample_u8 == Pillow's
    crop().resize(): full tiles, partial tiles (sizes that are not multiples of 32), windows at
    every image border, per-image and shared tables, a mild down-scale."""
    from leaffliction_amd import ops
    from leaffliction_amd.preprocessing import geometry as G
    n = 5
    x = batch_inputs(n, h, w, 31)
    xd = dev(x, cuda)
    rng = np.random.RandomState(h * 1000 + w)
    boxes = []
    for i in range(n):
        r = rng.uniform(0.8, 0.95)
        nw, nh = max(1, int(w * r)), max(1, int(h * r))
        left, top = [(0, 0), (w - nw, h - nh), (0, h - nh), (w - nw, 0)][i % 4] if i < 4 else \
            (rng.randint(0, w - nw + 1), rng.randint(0, h - nh + 1))
        boxes.append((left, top, nw, nh))
    t = ops.crop_resize_plan(w, h, boxes, xd.device)
    assert t[4] is True
    fused = ops.resample_u8(xd, h, w, t[0], t[1], t[2], t[3], True, tile_ok=True).cpu().numpy()
    two = ops.resample_u8(xd, h, w, t[0], t[1], t[2], t[3], True, tile_ok=False).cpu().numpy()
    assert np.array_equal(fused, two)
    for i in range(n):
        assert np.array_equal(fused[i], P.crop_resize_lanczos(x[i], *boxes[i])), i
    # shared tables: a mild down-scale that still fits the tile limits (8 taps need scale <= 7/6)
    oh, ow = (h * 9 // 10) // 4 * 4, (w * 9 // 10) // 4 * 4
    xb, xk, kx = G.lanczos_coeffs(w, 0.0, float(w), ow)
    yb, yk, ky = G.lanczos_coeffs(h, 0.0, float(h), oh)
    if ops.resample_tables_fit_tile(xb, xk, yb, yk, ow):
        tt = [dev(a, cuda) for a in (xb, xk, yb, yk)]
        a = ops.resample_u8(xd, oh, ow, tt[0], tt[1], tt[2], tt[3], False, tile_ok=True).cpu().numpy()
        b = ops.resample_u8(xd, oh, ow, tt[0], tt[1], tt[2], tt[3], False, tile_ok=False).cpu().numpy()
        assert np.array_equal(a, b)
    # tables the tile kernel must refuse: 13 taps (a 2x down-scale), width not a multiple of 4
    xb13, xk13, k13 = G.lanczos_coeffs(256, 0.0, 256.0, 128)
    assert k13 == 13 and not ops.resample_tables_fit_tile(xb13, xk13, xb13, xk13, 128)
    assert not ops.resample_tables_fit_tile(xb, xk, yb, yk, ow + 1)


def test_loader_resize_takes_the_fused_kernel(cuda):
    """ImageTransforms.resize_image 256 -> 224 (9 taps, the dataset's native size): the fused
    10-tap tile kernel equals the two-pass kernels and Pillow."""
    from leaffliction_amd import ops
    from leaffliction_amd.preprocessing import geometry as G
    x = batch_inputs(3, 256, 256, 41)
    xd = dev(x, cuda)
    xb, xk, kx = G.lanczos_coeffs(256, 0.0, 256.0, 224)
    assert kx == 9 and ops.resample_tables_fit_tile(xb, xk, xb, xk, 224)
    t = [dev(a, cuda) for a in (xb, xk)]
    two = ops.resample_u8(xd, 224, 224, t[0], t[1], t[0], t[1], False, tile_ok=False).cpu().numpy()
    got = ops.resize_lanczos_u8(xd, 224).cpu().numpy()
    assert np.array_equal(got, two)
    for i in range(3):
        assert np.array_equal(got[i], P.resize_lanczos(x[i], 224, 224))


def test_image_augmenter_file_interface(cuda, tmp_path):
    """ImageAugmenter(seed).<op>(src, dst) -> bool; failures return False, never raise."""
    from PIL import Image
    from leaffliction_amd.preprocessing.image_augmenter import ImageAugmenter, TRANSFORMATIONS
    src = tmp_path / "leaf.jpg"
    Image.fromarray(leaf_like(96, 96, 3)).save(src, quality=95)
    for op in TRANSFORMATIONS:
        assert getattr(ImageAugmenter(seed=42), op)(str(src), str(tmp_path / f"{op}.jpg")) is True
        assert (tmp_path / f"{op}.jpg").exists()
    assert ImageAugmenter(seed=1).flip(str(tmp_path / "missing.jpg"), str(tmp_path / "o.jpg")) is False
    png = tmp_path / "x.png"
    Image.fromarray(leaf_like(8, 8, 1)).save(png)
    assert ImageAugmenter(seed=1).rotate(str(png), str(tmp_path / "o.jpg")) is False  # .jpg only


def test_full_size_batch_properties(cuda):
    """BASELINE-size batch (256 x 224x224): properties that need no oracle pass."""
    import torch
    from leaffliction_amd import ops
    g = torch.Generator(device="cpu").manual_seed(42)
    x = torch.randint(0, 256, (256, 224, 224, 3), dtype=torch.uint8, generator=g).to(cuda)
    hist = ops.hist_u8(x)
    assert (hist.sum(-1) == 224 * 224).all()
    ref = torch.stack([torch.bincount(x[7, ..., c].flatten().long(), minlength=256)
                       for c in range(3)])
    assert torch.equal(hist[7].long(), ref)
    packed = ops.pack_hwc_u8_to_nchw_f32(x)
    assert torch.equal((packed * 255).round().to(torch.uint8), x.permute(0, 3, 1, 2))
    ident = torch.arange(256, dtype=torch.uint8, device=cuda).repeat(256, 3, 1)
    assert torch.equal(ops.lut_apply_u8(x, ident), x)
    m1 = torch.ones(256, dtype=torch.int32, device=cuda)
    assert torch.equal(ops.flip_u8(ops.flip_u8(x, m1), m1), x)


def test_gathers_stay_inside_a_batch_that_ends_its_allocation(cuda):
    """The bicubic warps fetch a footprint row with one unaligned 12-byte load and the nearest
    rotate fetches a pixel with one unaligned 4-byte load (lf_geom.hip: `x + 3 < w`, `sp < last`
    decide when the wide load is allowed).  A development build of round 1 without those guards read
    up to 9 bytes past the last pixel of the last image and aborted the process when the batch
    happened to end its allocation (gpurun_out/t10.log of round 1).  Here the
    batch is the FINAL slice of an exactly-sized allocation whose size is a multiple of the 2 MiB
    allocation granule, so such a read leaves the allocation; results must equal those of the same
    batch in the middle of a larger buffer."""
    import torch
    from leaffliction_amd import ops
    n, h, w = 4, 96, 128                                    # 147,456 bytes: a multiple of 16
    nbytes = n * h * w * 3
    big = torch.empty(2 << 20, dtype=torch.uint8, device=cuda)      # one whole 2 MiB granule
    g = torch.Generator().manual_seed(4)
    host = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g)
    tail = big[-nbytes:].view(n, h, w, 3)
    tail.copy_(host)
    mid = torch.empty(nbytes + 8192, dtype=torch.uint8, device=cuda)[4096:4096 + nbytes].view(n, h, w, 3)
    mid.copy_(host)
    f = [0.05, 0.08, 0.12, 0.15]
    skew = torch.tensor([[1 + v, 0, -v * w, 0, 1 + v, -v * h, 0, 0] for v in f], dtype=torch.float64, device=cuda)
    shear = torch.tensor([[1, 0.2, 0, 0, 1, 0, 0, 0], [1, 0, 0, -0.2, 1, 0, 0, 0], [1, -0.15, 0, 0, 1, 0, 0, 0],
                          [1, 0, 0, 0.1, 1, 0, 0, 0]], dtype=torch.float64, device=cuda)
    angles = [-30.0, 29.5, 0.0, 13.0]
    for a, b in ((ops.warp_bicubic_u8(tail, skew, True, True), ops.warp_bicubic_u8(mid, skew, True, True)),
                 (ops.warp_bicubic_u8(tail, shear, False), ops.warp_bicubic_u8(mid, shear, False))):
        assert torch.equal(a, b)
    for a, b in zip(ops.rotate_expand_u8(tail, angles), ops.rotate_expand_u8(mid, angles)):
        assert torch.equal(a, b)
    boxes = [(3, 2, 110, 80), (0, 0, 102, 77), (18, 16, 110, 80), (5, 9, 120, 86)]
    assert torch.equal(ops.crop_resize_lanczos_u8(tail, boxes), ops.crop_resize_lanczos_u8(mid, boxes))
    torch.cuda.synchronize()


@pytest.mark.parametrize("count", [1, 2, 7, 1000, 224 * 224 * 3])
def test_legacy_normal_planes_are_numpys(cuda, count):
    """ops.legacy_normal_u8 == np.random.RandomState(seed).normal(0, 5, n).astype(np.uint8) (image_augmenter.py:121-123),
    byte for byte, for every plane the kernel does not flag (a flag = a value within 1e-9 of an integer: the caller has
    that plane made on the host); flags are rare."""
    import torch
    from leaffliction_amd import ops
    seeds = [0, 1, 42, 123456, 999999, 2 ** 31, 2 ** 32 - 1] + list(range(1000, 1030))
    planes, flags = ops.legacy_normal_u8(seeds, 0.0, 5.0, count, cuda)
    planes, flags = planes.cpu().numpy(), flags.cpu().numpy()
    assert planes.shape == (len(seeds), count)
    assert (flags != 0).sum() <= 1 and not (flags == 2).any(), flags
    for i, sd in enumerate(seeds):
        if flags[i] == 0:
            want = np.random.RandomState(sd).normal(0, 5, count).astype(np.uint8)
            assert np.array_equal(planes[i], want), (sd, np.flatnonzero(planes[i] != want)[:5])
    # another location / scale
    p2, f2 = ops.legacy_normal_u8([7, 8], 100.0, 20.0, 5000, cuda)
    for i, sd in enumerate([7, 8]):
        if int(f2[i]) == 0:
            assert np.array_equal(p2[i].cpu().numpy(), np.random.RandomState(sd).normal(100.0, 20.0, 5000).astype(np.uint8))
    with pytest.raises(ValueError):
        ops.legacy_normal_u8([2 ** 32], 0.0, 5.0, 10, cuda)
