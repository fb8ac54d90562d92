"""Size-independent properties at BASELINE.json's full sizes (img 224, batch 256, fp32), where the
CPU oracle would take minutes: exact linearity under power-of-two scaling, determinism of the
slab-reduced weight gradient, epilogue statistics vs the two-pass kernels, histogram mass,
involutions of the byte kernels, and a full training step that stays finite and reproducible."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, S = 256, 224


def test_conv_fullsize_linearity_and_fused_statistics(cuda):
    from leaffliction_amd import nn
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(N, 32, S, S, generator=g).to(cuda)
    w = (torch.randn(32, 9, 32, generator=g) * 0.06).to(cuda)
    y = nn.conv2d(x, w, 3)
    # scaling by a power of two is exact in fp32: bit-identical results
    assert torch.equal(nn.conv2d(x * 2.0, w, 3), y * 2.0)
    assert torch.equal(nn.conv2d(x, w * 0.5, 3), y * 0.5)
    # zero padding: an all-ones image and kernel count the taps that fall inside the image
    ones = nn.conv2d(torch.ones(1, 32, S, S, device=cuda), torch.ones(32, 9, 32, device=cuda), 3)
    assert ones[0, 0, 0, 0].item() == 32 * 4 and ones[0, 5, 0, 7].item() == 32 * 6
    assert ones[0, 31, 100, 100].item() == 32 * 9
    # BatchNorm statistics from the conv epilogue == the two-pass statistics kernel
    gamma, beta = torch.ones(32, device=cuda), torch.zeros(32, device=cuda)
    st1, st2 = torch.zeros(4, 32, device=cuda), torch.zeros(4, 32, device=cuda)
    mm1, mv1 = torch.zeros(32, device=cuda), torch.ones(32, device=cuda)
    mm2, mv2 = torch.zeros(32, device=cuda), torch.ones(32, device=cuda)
    y1 = nn.conv2d_bn_stats(x, w, 3, gamma, beta, mm1, mv1, st1)
    assert torch.equal(y1, y)
    nn.bn_train_stats(y, gamma, beta, mm2, mv2, st2)
    assert torch.allclose(st1, st2, rtol=2e-5, atol=2e-6)
    ref_mean = y.double().mean((0, 2, 3)).float()
    assert torch.allclose(st1[0], ref_mean, atol=2e-6)


def test_wgrad_fullsize_is_deterministic_and_linear(cuda):
    from leaffliction_amd import nn
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(N, 32, S, S, generator=g).to(cuda)
    dy = torch.randn(N, 32, S, S, generator=g).to(cuda)
    a = nn.conv2d_wgrad(x, dy, 3)
    b = nn.conv2d_wgrad(x, dy, 3)
    assert torch.equal(a, b)                       # fixed-order slab reduction: no run-to-run drift
    assert torch.equal(nn.conv2d_wgrad(x, dy * 4.0, 3), a * 4.0)
    # sum over output channels of dW against an all-ones upstream = tap-shifted sums of x
    ones = torch.ones(N, 32, S, S, device=cuda)
    dw1 = nn.conv2d_wgrad(x, ones, 3)              # [32, 9, 32]
    centre = x.double().sum((0, 2, 3)).float()     # centre tap sees every pixel
    assert torch.allclose(dw1[:, 4, 0], centre, rtol=1e-4, atol=1e-2)


def test_augmentation_kernels_fullsize_properties(cuda):
    from leaffliction_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    n = 1024
    x = torch.randint(0, 256, (n, S, S, 3), dtype=torch.uint8, generator=g).to(cuda)
    h = ops.hist_u8(x)
    assert h.dtype == torch.int32 and bool((h.sum(-1) == S * S).all())      # every pixel lands in a bin
    assert torch.equal(h.sum(0).cpu(), torch.stack([torch.bincount(x[..., c].flatten().cpu().long(), minlength=256)
                                                     for c in range(3)]).int())
    for mode in (0, 1):                                                      # flips are involutions
        m = torch.full((n,), mode, dtype=torch.int32, device=cuda)
        assert torch.equal(ops.flip_u8(ops.flip_u8(x, m), m), x)
    p = ops.pack_hwc_u8_to_nchw_f32(x[:64])
    assert torch.equal((p * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1), x[:64])
    mask = torch.full((n, S, S), 255, dtype=torch.uint8, device=cuda)
    assert torch.equal(ops.mask_composite_u8(x, mask), x)                    # full mask keeps the image
    assert bool((ops.mask_composite_u8(x, torch.zeros_like(mask), "black") == 0).all())
    const = torch.full((4, S, S, 3), 77, dtype=torch.uint8, device=cuda)
    assert torch.equal(ops.gauss_blur_u8(const, 15, 0.0), const)             # taps sum to 256 exactly
    idx = torch.arange(n - 1, -1, -1, dtype=torch.int32, device=cuda)
    assert torch.equal(ops.gather_images_u8(x, idx), x.flip(0))


def test_training_step_fullsize_reproducible(cuda):
    from leaffliction_amd.model.cnn import LeafCNN
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randint(0, 256, (N, S, S, 3), dtype=torch.uint8, generator=g).to(cuda)
    y = torch.nn.functional.one_hot(torch.randint(0, 8, (N,), generator=g), 8).float().to(cuda)
    outs = []
    for _ in range(2):
        m = LeafCNN(num_classes=8, img_size=S, widths=(32, 64, 128, 256), drop_block=0.15, drop_top=0.4,
                    l2_reg=1e-4, augment=True, use_se=True, seed=9, device=cuda)
        losses = []
        for step in range(3):
            _p, loss = m.train_step(x, y, 1e-3)
            losses.append(loss.mean().item())
        outs.append((losses, m.flat_p.clone(), m.flat_g.clone()))
    (l1, p1, g1), (l2, p2, g2) = outs
    assert all(np.isfinite(l1)) and l1 == l2                 # same seeds -> same bits, step after step
    assert torch.equal(p1, p2) and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0
    assert 0.5 < l1[0] < 8.0                                 # random init: a few nats, not an overflow


def test_blur_mask_jpeg_fullsize_properties(cuda):
    """Round-2 kernels at batch size: the i8-MFMA blur commutes with mirror images (symmetric taps, reflect-101
    borders) and equals the dot-product kernels where both apply; the slotted histogram equals the per-wave one;
    the JPEG halves invert each other the way libjpeg's do (decode(encode(x)) is what Pillow's round trip gives)."""
    import io

    from PIL import Image

    from leaffliction_amd import ops
    from leaffliction_amd.utils import jpeg_host
    g = torch.Generator(device="cpu").manual_seed(11)
    n = 768
    x = torch.randint(0, 256, (n, S, S, 3), dtype=torch.uint8, generator=g).to(cuda)
    for k in (5, 15):
        b = ops.gauss_blur_u8(x, k, 0.0)
        assert torch.equal(ops.gauss_blur_u8(x.flip(1), k, 0.0), b.flip(1))      # upside down
        assert torch.equal(ops.gauss_blur_u8(x.flip(2), k, 0.0), b.flip(2))      # mirrored
        # a ragged crop goes to the dot-product kernel; its interior (far from the crop's borders) must agree
        c = ops.gauss_blur_u8(x[:8, :220, :219].contiguous(), k, 0.0)
        assert torch.equal(c[:, 16:200, 16:200], b[:8, 16:200, 16:200])
    big, small = ops.hist_u8(x), torch.cat([ops.hist_u8(x[i:i + 256]) for i in range(0, n, 256)])
    assert torch.equal(big, small)                                               # >= 512 images: slotted table
    coef = ops.jpeg_fdct_quant_u8(x[:512])
    rows = ops.jpeg_entropy_u8(coef, S, S).cpu().numpy()
    lens = rows[:, :4].copy().view(np.int32)[:, 0]
    assert (lens > 0).all()
    files = [jpeg_host.wrap_scan(rows[i, 4:4 + lens[i]], S, S) for i in (0, 255, 511)]
    stride = (256 + 3 * S * S + 4095) // 4096 * 4096
    slots = np.zeros((3, stride), np.uint8)
    for i, f in enumerate(files):
        assert jpeg_host.read_file_into(f, slots[i]) == (S, S)
    back = ops.jpeg_idct_rgb_u8(torch.from_numpy(slots).to(cuda), S, S).cpu().numpy()
    xs = x[[0, 255, 511]].cpu().numpy()
    for i, f in enumerate(files):
        b = io.BytesIO()
        Image.fromarray(xs[i]).save(b, format="JPEG", quality=95)
        assert b.getvalue() == f                                                  # Pillow's file
        assert np.array_equal(back[i], np.asarray(Image.open(io.BytesIO(f)).convert("RGB")))   # Pillow's pixels
