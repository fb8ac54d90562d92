"""Known-answer tests that guard the UNPINNED halves of the oracle (OpenCV 8-bit conventions
and the Keras layer/optimizer semantics restated in oracle/cv_ops.py and oracle/cnn_ref.py).
The list follows SURVEY §8c "Build-side KATs to create"."""
import math

import numpy as np
import pytest
import torch

from oracle import cnn_ref as R
from oracle import cv_ops as CV


# ------------------------------------------------------------------ OpenCV conventions
def test_hsv_cube_corners_and_greys():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255],
                    [255, 0, 255], [0, 0, 0], [255, 255, 255], [128, 128, 128], [17, 17, 17]]],
                  dtype=np.uint8)
    hsv = CV.rgb2hsv(px)[0].tolist()
    assert hsv[0] == [0, 255, 255]      # red
    assert hsv[1] == [60, 255, 255]     # green  (H in [0,180): 120 deg / 2)
    assert hsv[2] == [120, 255, 255]    # blue
    assert hsv[3] == [30, 255, 255]     # yellow
    assert hsv[4] == [90, 255, 255]     # cyan
    assert hsv[5] == [150, 255, 255]    # magenta
    for g in hsv[6:]:                   # greys: H = 0, S = 0, V = value
        assert g[0] == 0 and g[1] == 0
    assert [g[2] for g in hsv[6:]] == [0, 255, 128, 17]
    assert CV.rgb2hsv(np.random.RandomState(0).randint(0, 256, (50, 50, 3)).astype(np.uint8))[..., 0].max() < 180


def test_gray_coefficients():
    assert 4899 + 9617 + 1868 == 1 << 14           # weights sum to one in Q14
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    assert CV.rgb2gray(px)[0].tolist() == [255, 0, 76, 150, 29]   # 0.299 / 0.587 / 0.114


def test_gaussian_kernel_rules():
    for k, s in [(15, 0.0), (5, 1.5), (3, 0.0), (7, 2.0)]:
        q = CV.gaussian_kernel_q8(k, s)
        assert q.sum() == 256 and (q == q[::-1]).all() and q.argmax() == k // 2
    # sigma=0 rule: 0.3*((k-1)*0.5 - 1) + 0.8 -> 2.6 for k = 15 (blur.py:61)
    assert abs(0.3 * ((15 - 1) * 0.5 - 1) + 0.8 - 2.6) < 1e-12
    q0 = CV.gaussian_kernel_q8(15, 0.0)
    q26 = CV.gaussian_kernel_q8(15, 2.6)
    assert (q0 == q26).all()


def test_blur_reflect101_and_constants():
    flat = np.full((9, 11, 3), 93, np.uint8)
    assert (CV.gaussian_blur(flat, 15, 0.0) == 93).all()           # taps sum to 1: constants stay
    ramp = np.tile(np.arange(0, 80, 4, dtype=np.uint8), (6, 1))     # linear ramp in x
    out = CV.gaussian_blur(ramp, 5, 1.5)
    assert (out[:, 2:-2] == ramp[:, 2:-2]).all()                    # symmetric kernel keeps a ramp
    # BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba): the edge pixel is not duplicated
    assert CV._reflect101(np.array([-1, -2, 5, 6]), 5).tolist() == [1, 2, 3, 2]


def test_apply_mask_docstring_example():
    """srcs/utils/mask_utils.py:29-41."""
    rng = np.random.RandomState(1)
    img = rng.randint(0, 255, (100, 100, 3)).astype(np.uint8)
    mask = rng.randint(0, 2, (100, 100)).astype(np.uint8) * 255
    w, b = CV.apply_mask(img, mask, "white"), CV.apply_mask(img, mask, "black")
    assert (w[mask == 0] == 255).all() and (b[mask == 0] == 0).all()
    assert (w[mask == 255] == img[mask == 255]).all()
    with pytest.raises(ValueError):
        CV.apply_mask(img, mask, "green")


def test_region_stats_consistency():
    img = np.zeros((4, 4, 3), np.uint8)
    img[:2] = (60, 140, 50)     # healthy green
    img[2:] = (120, 70, 30)     # brown
    c, hist = CV.hsv_region_stats(img)
    assert c[0] == 16 and c[1] == 8 and c[4] == 8     # leaf, Vert Sain, Brun/Orange
    assert hist.sum(axis=1).tolist() == [16, 16, 16]


# ------------------------------------------------------------------ Keras layer semantics
def test_conv_delta_kernel_is_identity():
    x = torch.randn(2, 3, 5, 7)
    w = torch.zeros(3, 9, 3)
    for c in range(3):
        w[c, 4, c] = 1.0
    assert torch.equal(R.conv(x, w, 3), x)


def test_batchnorm_hand_moments():
    y = torch.tensor([1.0, 3.0, 5.0, 7.0]).view(4, 1, 1, 1)        # mean 4, biased var 5
    st = {"bn.mean": torch.zeros(1), "bn.var": torch.ones(1)}
    out = R.batchnorm(y, torch.tensor([2.0]), torch.tensor([0.5]), st, "bn", True)
    exp = (y - 4.0) / math.sqrt(5.0 + 1e-3) * 2.0 + 0.5
    assert torch.allclose(out, exp, atol=1e-6)
    assert st["bn.mean"].item() == pytest.approx(0.04)              # 0*0.99 + 4*0.01
    assert st["bn.var"].item() == pytest.approx(0.99 + 0.05)        # biased variance, no Bessel
    inf = R.batchnorm(y, torch.tensor([1.0]), torch.tensor([0.0]), st, "bn", False)
    assert torch.allclose(inf, (y - 0.04) / math.sqrt(1.04 + 1e-3), atol=1e-6)


def test_se_zero_weights_scale_by_half():
    widths, classes = [16], 2
    p = R.init_params(classes, widths, seed=1)
    for k in ("s0.se.w1", "s0.se.w2"):
        p[k].zero_()
    st = R.init_state(widths)
    got = {}
    R.forward(p, st, torch.randn(2, 3, 8, 8), widths, False, collect=got)
    p2 = {k: v.clone() for k, v in p.items()}
    p2["s0.se.b2"] = torch.full_like(p["s0.se.b2"], 50.0)          # sigmoid -> 1
    got2 = {}
    R.forward(p2, R.init_state(widths), torch.randn(2, 3, 8, 8), widths, False, collect=got2)
    # with zero SE weights the gate is sigmoid(0) = 0.5 for every channel
    z = torch.sigmoid(torch.zeros(1))
    assert z.item() == 0.5


def test_label_smoothed_cce_uniform_is_log_c():
    c = 8
    probs = torch.full((3, c), 1.0 / c)
    y = R.smooth_labels(torch.nn.functional.one_hot(torch.tensor([0, 3, 7]), c).float(), 0.02)
    assert torch.allclose(y.sum(-1), torch.ones(3))
    assert y[0, 0].item() == pytest.approx(0.98 + 0.02 / 8)
    assert torch.allclose(R.cce_loss(probs, y), torch.full((3,), math.log(c)), atol=1e-6)


def test_adamw_one_step_closed_form():
    w0 = torch.tensor([1.0, -2.0])
    g = torch.tensor([0.3, 0.4])                                   # norm 0.5: at the clip edge
    p, m, v = R.adamw_step({"w": w0.clone()}, {"w": g}, {"w": torch.zeros(2)}, {"w": torch.zeros(2)},
                           step=1, lr=1e-2, wd=1e-4, clipnorm=0.5)
    # first step: m_hat = g, v_hat = g^2 -> update = lr * g/(|g| + eps*...) ~ lr * sign(g)
    exp = w0 - w0 * 1e-4 * 1e-2 - 1e-2 * torch.sign(g)
    assert torch.allclose(p["w"], exp, atol=1e-6)
    g_big = torch.tensor([3.0, 4.0])                               # norm 5 -> scaled to 0.5
    p2, m2, _ = R.adamw_step({"w": w0.clone()}, {"w": g_big}, {"w": torch.zeros(2)},
                             {"w": torch.zeros(2)}, step=1, lr=1e-2, clipnorm=0.5)
    assert torch.allclose(m2["w"], torch.tensor([0.3, 0.4]) * 0.1, atol=1e-7)


def test_maxpool_and_gap_on_ramps():
    x = torch.arange(16.0).view(1, 1, 4, 4)
    assert torch.nn.functional.max_pool2d(x, 2).flatten().tolist() == [5.0, 7.0, 13.0, 15.0]
    assert x.mean(dim=(2, 3)).item() == 7.5


def test_param_count_matches_survey():
    """base preset, 8 classes -> 1,250,756 trainable parameters."""
    n = sum(int(np.prod(s)) for _n, s, _k in R.param_specs(8, [32, 64, 128, 256]))
    assert n == 1_250_756
    assert sum(int(np.prod(s)) for _n, s, _k in R.param_specs(8, [32, 64, 128])) == 314_020
    assert sum(int(np.prod(s)) for _n, s, _k in R.param_specs(8, [16, 32, 64])) == 79_382


def test_canny_step_edge_and_flat():
    """A vertical 0->255 step gives equal Sobel magnitudes on both sides of the step; Canny's
    asymmetric test (m > left, m >= right) keeps the left one only.  Flat input: no edges."""
    g = np.zeros((8, 10), np.uint8)
    g[:, 5:] = 255
    e = CV.canny(g, 50, 150)
    assert e[:, 4].tolist() == [255] * 8 and int(e.sum()) == 8 * 255
    assert int(CV.canny(g.T.copy(), 50, 150)[4].sum()) == 8 * 255      # horizontal step: row 4
    assert int(CV.canny(np.full((6, 6), 77, np.uint8), 50, 150).sum()) == 0


def test_canny_hysteresis_keeps_weak_only_next_to_strong():
    """A ramp edge whose gradient magnitude is between the thresholds (weak) survives only where it
    touches a strong segment."""
    g = np.zeros((12, 12), np.uint8)
    g[:6, 6:] = 200          # strong edge on rows 0..5 (dx = 800 > 150)
    g[6:, 6:] = 20           # weak edge on rows 6..11 (dx = 80, between 50 and 150)
    e = CV.canny(g, 50, 150)
    assert e[:5, 5].tolist() == [255] * 5
    assert e[6:, 5].tolist() == [255] * 6      # weak run connected to the strong one: kept
    iso = np.zeros((12, 12), np.uint8)
    iso[:, 6:] = 20          # weak everywhere, no strong seed: dropped
    assert int(CV.canny(iso, 50, 150).sum()) == 0


def test_morph_cross_and_minmax_normalise():
    m = np.zeros((5, 5), np.uint8)
    m[2, 2] = 255
    d = CV.morph_cross3(m, erode=False)
    assert d.tolist() == [[0, 0, 0, 0, 0], [0, 0, 255, 0, 0], [0, 255, 255, 255, 0], [0, 0, 255, 0, 0],
                          [0, 0, 0, 0, 0]]
    assert int(CV.morph_cross3(d, erode=True).sum()) == 255 and CV.morph_cross3(d, erode=True)[2, 2] == 255
    full = np.full((4, 4), 255, np.uint8)     # the border never erodes a full image
    assert np.array_equal(CV.morph_cross3(full, erode=True), full)
    x = np.array([[0.0, 1.0], [2.0, 3.0]], np.float32)
    assert CV.normalize_minmax_f32(x).tolist() == [[0.0, 85.0], [170.0, 255.0]]
    assert CV.normalize_minmax_f32(np.full((3, 3), 7.5, np.float32)).tolist() == [[0.0] * 3] * 3


def test_blur_saliency_flat_image_and_mask():
    """Flat image: no edges, zero gradient, zero colour difference -> all-zero output; the output
    is gray replicated to RGB and zero outside the leaf mask."""
    flat = np.full((20, 24, 3), 90, np.uint8)
    mask = np.full((20, 24), 255, np.uint8)
    assert int(CV.blur_saliency(flat, mask).sum()) == 0
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (20, 24, 3)).astype(np.uint8)
    mask[:, :12] = 0
    out = CV.blur_saliency(img, mask)
    assert out.shape == (20, 24, 3) and int(out[:, :12].sum()) == 0 and int(out[:, 12:].sum()) > 0
    assert np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 0], out[..., 2])


# ---- _create_inclusive_mask (mask.py:727-831): L*a*b*, ellipse elements, morphology, components -------------
def test_lab_primaries_and_greys():
    """cv2.cvtColor(.., COLOR_RGB2LAB) on 8-bit input: the values OpenCV's documentation and every cv2 session
    give for the sRGB primaries, white, black and mid grey."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)
    assert CV.rgb2lab(px)[0].tolist() == [[136, 208, 195], [224, 42, 211], [82, 207, 20], [255, 128, 128],
                                          [0, 128, 128], [137, 128, 128]]
    g = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    lab = CV.rgb2lab(g)[0]
    assert (lab[:, 1] == 128).all() and (lab[:, 2] == 128).all()          # greys have no chroma
    assert (np.diff(lab[:, 0].astype(int)) >= 0).all() and lab[0, 0] == 0 and lab[255, 0] == 255
    gam, cbrt = CV.lab_tables()
    assert gam[0] == 0 and gam[255] == 2040 and cbrt[2040] == 32768       # f(1) = 1 at 15 bits
    assert (np.diff(gam.astype(int)) >= 0).all() and (np.diff(cbrt.astype(int)) > 0).all()


def test_ellipse_elements():
    """cv2.getStructuringElement(MORPH_ELLIPSE, (k, k)): 3 is the plus, 5 loses its four corners and the
    rows next to them stay full, 7 and 9 as OpenCV prints them."""
    assert CV.ellipse_se(3).astype(int).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert CV.ellipse_se(5).astype(int).tolist() == [[0, 0, 1, 0, 0]] + [[1] * 5] * 3 + [[0, 0, 1, 0, 0]]
    assert CV.ellipse_se(7).sum(axis=1).tolist() == [1, 5, 7, 7, 7, 5, 1]
    assert CV.ellipse_se(9).sum(axis=1).tolist() == [1, 7, 7, 9, 9, 9, 7, 7, 1]
    for k in (3, 5, 7, 9):
        se = CV.ellipse_se(k)
        assert np.array_equal(se, se[::-1]) and np.array_equal(se, se[:, ::-1])   # mirror-symmetric, wider than tall
        assert np.array_equal(CV.morph(np.eye(1, dtype=np.uint8) * 255, se, False), [[255]])


def test_morphology_and_border():
    m = np.zeros((9, 9), np.uint8)
    m[4, 4] = 255
    d = CV.morph(m, CV.ellipse_se(5), erode=False)
    assert np.array_equal(d[2:7, 2:7] > 0, CV.ellipse_se(5))               # dilating a point draws the element
    assert np.array_equal(CV.morph(d, CV.ellipse_se(5), erode=True), m)    # and eroding brings the point back
    full = np.full((6, 7), 255, np.uint8)
    assert np.array_equal(CV.morph(full, CV.ellipse_se(9), erode=True), full)   # the border never erodes
    assert np.array_equal(CV.morph_cross3(m, False), CV.morph(m, CV.ellipse_se(3), False))
    hole = full.copy()
    hole[3, 3] = 0
    assert np.array_equal(CV.morph_close(hole, 3), full) and np.array_equal(CV.morph_open(m, 3), np.zeros_like(m))


def test_largest_component_rules():
    m = np.zeros((6, 8), np.uint8)
    m[0, 0:2] = 255            # area 2
    m[2, 2] = m[3, 3] = m[4, 2] = 255   # diagonal neighbours join under 8-connectivity: area 3
    m[5, 6:8] = 255            # area 2
    out = CV.largest_component(m)
    assert int(out.sum()) == 3 * 255 and out[3, 3] == 255 and out[0, 0] == 0
    m[3, 3] = 0                # now 2, 1, 1, 2: the first of the equal largest (raster order) stays
    out = CV.largest_component(m)
    assert int(out.sum()) == 2 * 255 and out[0, 0] == 255 and out[5, 7] == 0
    assert np.array_equal(CV.largest_component(np.zeros((3, 3), np.uint8)), np.zeros((3, 3), np.uint8))


def test_canny_l1_thresholds():
    """cv2.Canny's default gradient: |dx| + |dy| against floor(threshold).  A step of 25 has |dx| = 100 on its
    two columns: strong at high = 99, nothing at all at low = high = 100."""
    g = np.zeros((8, 10), np.uint8)
    g[:, 5:] = 25
    assert int(CV.canny(g, 30, 99, l2gradient=False).sum()) == 8 * 255
    assert int(CV.canny(g, 100, 100, l2gradient=False).sum()) == 0


def test_inclusive_mask_on_a_synthetic_leaf():
    """A green disc on a flat grey card: the mask is the disc (to within the closes), nothing of the card; a
    second, smaller green blob is dropped by the largest-component rule; an all-grey image gives no mask."""
    h = w = 96
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w, 3), 128, np.uint8)
    disc = (yy - 48) ** 2 + (xx - 40) ** 2 <= 24 ** 2
    blob = (yy - 12) ** 2 + (xx - 84) ** 2 <= 5 ** 2
    img[disc] = (40, 160, 50)
    img[blob] = (40, 160, 50)
    m = CV.inclusive_mask(img)
    assert set(np.unique(m)) <= {0, 255}
    inner = (yy - 48) ** 2 + (xx - 40) ** 2 <= 21 ** 2
    outer = (yy - 48) ** 2 + (xx - 40) ** 2 <= 29 ** 2
    assert (m[inner] == 255).all() and (m[~outer] == 0).all() and (m[blob] == 0).all()
    assert int(CV.inclusive_mask(np.full((40, 40, 3), 128, np.uint8)).sum()) == 0
