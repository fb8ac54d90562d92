"""lf_inclusive_mask_u8 (the default make_mask strategy's candidate mask, mask.py:727-831) against
oracle/cv_ops.py:inclusive_mask: every pixel of the 0 / 255 plane, on leaf-like scenes, on noise (thousands of
tiny components through the morphology and the union-find), on ragged sizes and on degenerate planes."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import cv_ops as CV  # noqa: E402

pytestmark = pytest.mark.gpu


def leaf_scene(h, w, seed):
    """A textured green blob with brown spots on a grey / purple card, a second smaller blob, sensor noise."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cy, cx = h * rng.uniform(0.4, 0.6), w * rng.uniform(0.4, 0.6)
    ang = rng.uniform(0, np.pi)
    u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
    v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
    leaf = (u / (0.36 * w)) ** 2 + (v / (0.22 * h)) ** 2 <= 1.0 + 0.08 * np.sin(7 * np.arctan2(v, u))
    bg = np.array([[120, 118, 125], [112, 100, 128]][seed % 2], np.float64)
    img = np.ones((h, w, 3)) * bg + rng.normal(0, 2.0 + seed % 3, (h, w, 3))
    green = np.array([50, 140 + 10 * (seed % 4), 45], np.float64)
    tex = 12 * np.sin(u / 3.0) * np.cos(v / 5.0)
    img[leaf] = green + tex[leaf, None] + rng.normal(0, 4, (int(leaf.sum()), 3))
    for _ in range(4):
        sy, sx = cy + rng.uniform(-0.1, 0.1) * h, cx + rng.uniform(-0.2, 0.2) * w
        spot = (yy - sy) ** 2 + (xx - sx) ** 2 <= (0.03 * w) ** 2
        img[spot & leaf] = np.array([120, 80, 40]) + rng.normal(0, 5, (int((spot & leaf).sum()), 3))
    small = (yy - 0.1 * h) ** 2 + (xx - 0.88 * w) ** 2 <= (0.04 * w) ** 2
    img[small] = green
    return np.clip(img, 0, 255).astype(np.uint8)


def run(batch, cuda, green=(25, 100)):
    from leaffliction_amd import ops
    got = ops.inclusive_mask_u8(torch.from_numpy(np.ascontiguousarray(batch)).to(cuda), green).cpu().numpy()
    for i in range(batch.shape[0]):
        want = CV.inclusive_mask(batch[i], green)
        assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()))
    return got


@pytest.mark.parametrize("h,w", [(224, 224), (96, 130), (291, 291), (33, 17), (64, 64)])
def test_inclusive_mask_leaf_scenes(cuda, h, w):
    got = run(np.stack([leaf_scene(h, w, s) for s in range(4)]), cuda)
    if min(h, w) >= 64:
        assert all(0.05 < (g > 0).mean() < 0.6 for g in got)     # the scenes do produce a leaf-sized mask


def test_inclusive_mask_noise_and_flat_planes(cuda):
    rng = np.random.RandomState(5)
    noise = rng.randint(0, 256, (3, 80, 100, 3)).astype(np.uint8)
    run(noise, cuda)
    flat = np.stack([np.full((80, 100, 3), 128, np.uint8), np.zeros((80, 100, 3), np.uint8),
                     np.tile(np.array([40, 170, 50], np.uint8), (80, 100, 1))])
    got = run(flat, cuda)
    assert int(got[0].sum()) == 0 and (got[2] == 255).all()
    # salt of isolated green pixels and a checkerboard of 2x2 blocks: many components, equal areas
    img = np.full((2, 72, 96, 3), 128, np.uint8)
    img[0, ::6, ::6] = (40, 170, 50)
    yy, xx = np.mgrid[0:72, 0:96]
    img[1][((yy // 2 + xx // 2) % 4 == 0)] = (40, 170, 50)
    run(img, cuda)


def test_inclusive_mask_green_range_and_host_mirror(cuda):
    from leaffliction_amd.transform import filters as F
    scene = leaf_scene(128, 128, 3)
    run(scene[None], cuda, green=(40, 70))
    cfg = F.TransformConfig()
    assert np.array_equal(F.create_inclusive_mask(scene, cfg), CV.inclusive_mask(scene, cfg.green_hue_range))
