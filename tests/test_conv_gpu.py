"""conv2d forward / dgrad / wgrad (fp32 MFMA) vs torch CPU fp32 conv2d (the float oracle).

Tolerance: the kernel is an fp32 fmaf chain over K = Cin*9 (<= 2304) terms; torch's CPU
conv uses a different summation order, so compare at rtol 2e-4 of the output scale.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def iko_to_oihw(w_iko, k):
    cin, taps, cout = w_iko.shape
    return w_iko.permute(2, 0, 1).reshape(cout, cin, k, k).contiguous()


def close(got, ref, tol=2e-4):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


SHAPES = [  # n, cin, cout, h, w, k
    (2, 3, 32, 64, 64, 3),     # stem-like (Cin=3)
    (2, 32, 32, 64, 64, 3),
    (2, 32, 64, 32, 32, 3),
    (2, 64, 128, 56, 56, 3),   # 28x8 tiles
    (1, 128, 256, 28, 28, 3),  # masked 28-wide tiles, two cout tiles
    (3, 128, 256, 28, 28, 3),  # two-image strips (H = 28 is 3.5 tiles) + a lone last image
    (4, 64, 128, 28, 28, 1),   # 1x1 on strips
    (2, 16, 16, 20, 12, 3),    # ragged: Cout < 32, partial tiles
    (2, 5, 7, 9, 11, 3),       # everything ragged
    (2, 32, 64, 32, 32, 1),    # 1x1 projection
    (1, 128, 256, 28, 28, 1),
    (3, 8, 40, 17, 5, 1),
]


@pytest.mark.parametrize("n,cin,cout,h,w,k", SHAPES)
def test_conv_forward_dgrad_wgrad(cuda, n, cin, cout, h, w, k):
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(n * 1000 + cin * 10 + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, k * k, cout, generator=g) / (cin * k * k) ** 0.5
    dy = torch.randn(n, cout, h, w, generator=g)
    w_oihw = iko_to_oihw(wt, k)
    xr = x.clone().requires_grad_(True)
    wr = w_oihw.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, padding=k // 2)
    y_ref.backward(dy)
    xd, wd, dyd = x.to(cuda), wt.to(cuda), dy.to(cuda)
    close(nn.conv2d(xd, wd, k).cpu(), y_ref.detach())
    dx = nn.conv2d(dyd, nn.conv2d_dgrad_weights(wd, k), k).cpu()
    close(dx, xr.grad)
    dw = nn.conv2d_wgrad(xd, dyd, k).cpu()
    dw_ref = wr.grad.reshape(cout, cin, k * k).permute(1, 2, 0)
    close(dw, dw_ref, tol=5e-4)


def test_conv_fused_prologue(cuda):
    """Prologue = producer's BatchNorm+ReLU applied while staging; padding stays zero."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 24, 40, generator=g)
    wt = torch.randn(32, 9, 64, generator=g) / 17.0
    sc = torch.rand(32, generator=g) + 0.5
    sh = torch.randn(32, generator=g) * 0.3
    dy = torch.randn(2, 64, 24, 40, generator=g)
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ar = a.clone().requires_grad_(True)
    wr = iko_to_oihw(wt, 3).clone().requires_grad_(True)
    y_ref = F.conv2d(ar, wr, padding=1)
    y_ref.backward(dy)
    xd, wd, scd, shd, dyd = (t.to(cuda) for t in (x, wt, sc, sh, dy))
    close(nn.conv2d(xd, wd, 3, scd, shd, True).cpu(), y_ref.detach())
    dw = nn.conv2d_wgrad(xd, dyd, 3, scd, shd, True).cpu()
    close(dw, wr.grad.reshape(64, 32, 9).permute(1, 2, 0), tol=5e-4)


def test_conv_known_answers(cuda):
    """Delta kernel reproduces the input; ones kernel counts the valid neighbours (zero pad)."""
    from leaffliction_amd import nn
    x = torch.arange(2 * 4 * 6 * 6, dtype=torch.float32).reshape(2, 4, 6, 6).to(cuda)
    w = torch.zeros(4, 9, 4, device=cuda)
    for c in range(4):
        w[c, 4, c] = 1.0  # centre tap, identity over channels
    assert torch.equal(nn.conv2d(x, w, 3), x)
    ones = torch.ones(1, 1, 5, 5, device=cuda)
    cnt = nn.conv2d(ones, torch.ones(1, 9, 1, device=cuda), 3)[0, 0].cpu()
    assert cnt[0, 0] == 4 and cnt[0, 2] == 6 and cnt[2, 2] == 9
    # determinism: the slab reduce has a fixed order
    g = torch.Generator().manual_seed(1)
    a = torch.randn(4, 32, 32, 32, generator=g).to(cuda)
    d = torch.randn(4, 32, 32, 32, generator=g).to(cuda)
    assert torch.equal(nn.conv2d_wgrad(a, d, 3), nn.conv2d_wgrad(a, d, 3))


@pytest.mark.parametrize("n,cin,cout,h,w,k", SHAPES)
def test_conv_bn_stats_epilogue(cuda, n, cin, cout, h, w, k):
    """Conv2D + training BatchNormalization statistics gathered in the conv epilogue: same y
    (bit-identical to the plain conv), batch mean / biased variance, scale / shift and moving
    statistics vs torch fp32 (mean 1e-5, variance 1e-4 relative: fp32 tile sums about the
    moving-mean pivot, combined in double)."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(7 + n * 1000 + cin * 10 + cout + h)
    x = torch.randn(n, cin, h, w, generator=g) + 0.5
    wt = torch.randn(cin, k * k, cout, generator=g) / (cin * k * k) ** 0.5
    gamma = torch.rand(cout, generator=g) + 0.5
    beta = torch.randn(cout, generator=g)
    mmean0 = torch.randn(cout, generator=g) * 0.3
    mvar0 = torch.rand(cout, generator=g) + 0.5
    y_ref = F.conv2d(x, iko_to_oihw(wt, k), padding=k // 2)
    mean_ref = y_ref.mean((0, 2, 3))
    var_ref = y_ref.var((0, 2, 3), unbiased=False)
    xd, wd = x.to(cuda), wt.to(cuda)
    mm, mv = mmean0.to(cuda), mvar0.to(cuda)
    stats = torch.zeros(4, cout, device=cuda)
    y = nn.conv2d_bn_stats(xd, wd, k, gamma.to(cuda), beta.to(cuda), mm, mv, stats,
                           momentum=0.99, eps=1e-3)
    assert torch.equal(y, nn.conv2d(xd, wd, k))
    st = stats.cpu()
    scale = y_ref.abs().max().item()
    assert (st[0] - mean_ref).abs().max().item() <= 1e-5 * scale
    assert ((st[1] - 1.0 / torch.sqrt(var_ref + 1e-3)).abs() * torch.sqrt(var_ref + 1e-3)).max().item() <= 1e-4
    sc_ref = gamma / torch.sqrt(var_ref + 1e-3)
    assert torch.allclose(st[2], sc_ref, rtol=1e-4, atol=1e-6)
    assert torch.allclose(st[3], beta - mean_ref * sc_ref, rtol=1e-4, atol=1e-4 * scale)
    assert torch.allclose(mm.cpu(), mmean0 * 0.99 + mean_ref * 0.01, rtol=1e-5, atol=1e-6)
    assert torch.allclose(mv.cpu(), mvar0 * 0.99 + var_ref * 0.01, rtol=1e-5, atol=1e-6)
    # and against the two-pass path (plain conv, then the statistics kernel)
    mm2, mv2 = mmean0.to(cuda), mvar0.to(cuda)
    stats2 = torch.zeros(4, cout, device=cuda)
    nn.bn_train_stats(y, gamma.to(cuda), beta.to(cuda), mm2, mv2, stats2, 0.99, 1e-3)
    assert torch.allclose(stats, stats2, rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("n,cin,cout,h,w,with_se,k,relu", [
    (3, 32, 32, 32, 32, True, 3, True),     # fused, CI_T == Cin
    (2, 64, 64, 32, 32, False, 3, True),    # two ci blocks
    (2, 32, 64, 56, 56, True, 3, True),     # 28-wide tiles
    (2, 3, 32, 64, 64, False, 3, True),     # stem: small-Cin kernel
    (2, 32, 64, 32, 32, False, 1, False),   # 1x1 projection BN (no ReLU)
    (2, 64, 128, 56, 56, False, 1, False),
    (2, 16, 24, 20, 12, True, 3, True)])    # unsupported -> fallback
def test_bn_backward_inside_wgrad(cuda, n, cin, cout, h, w, with_se, k, relu):
    """BatchNorm backward formed inside the weight-gradient kernel (dy written on the side) vs
    the standalone BN-backward kernel followed by the plain wgrad, and vs torch autograd."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(99 + n + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    y = torch.randn(n, cout, h, w, generator=g) * 1.5 + 0.3          # BN input
    up = torch.randn(n, cout, h, w, generator=g)                      # upstream gradient
    gamma = torch.rand(cout, generator=g) + 0.5
    beta = torch.randn(cout, generator=g) * 0.2
    alpha = torch.rand(n, cout, generator=g) + 0.5 if with_se else None
    add = torch.randn(n, cout, generator=g) * 0.01 if with_se else None
    in_sc, in_sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1

    # torch reference: a = relu(BN(y)); L = sum(a * (up*alpha) ) + sum(a * add) -> dL/dy, then wgrad
    yr = y.clone().requires_grad_(True)
    mean = yr.mean((0, 2, 3), keepdim=True)
    var = yr.var((0, 2, 3), unbiased=False, keepdim=True)
    a = (yr - mean) / torch.sqrt(var + 1e-3) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    if relu:
        a = torch.relu(a)
    coeff = up * (alpha.view(n, cout, 1, 1) if with_se else 1.0) + (add.view(n, cout, 1, 1) if with_se else 0.0)
    (a * coeff).sum().backward()
    dy_ref = yr.grad
    xp = torch.relu(x * in_sc.view(1, -1, 1, 1) + in_sh.view(1, -1, 1, 1))
    wr = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(xp, wr, padding=k // 2).backward(dy_ref)
    dw_ref = wr.grad.reshape(cout, cin, k * k).permute(1, 2, 0)

    d = lambda t: None if t is None else t.to(cuda)  # noqa: E731
    stats = torch.zeros(4, cout, device=cuda)
    mm, mv = torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda)
    nn.bn_train_stats(d(y), d(gamma), d(beta), mm, mv, stats, 0.99, 1e-3)
    outs = []
    for fused in (True, False):
        dgamma, dbeta = torch.zeros(cout, device=cuda), torch.zeros(cout, device=cuda)
        dw = torch.zeros(cin, k * k, cout, device=cuda)
        dy = torch.zeros(n, cout, h, w, device=cuda)
        if fused:
            nn.bn_bwd_wgrad(d(x), d(up), d(y), stats, d(gamma), dgamma, dbeta, relu, k, dw, dy,
                            d(in_sc), d(in_sh), True, alpha_nc=d(alpha), add_nc=d(add))
        else:
            nn.bn_bwd(d(up), d(y), stats, d(gamma), dgamma, dbeta, relu, alpha_nc=d(alpha),
                      add_nc=d(add), out=dy)
            nn.conv2d_wgrad(d(x), dy, k, d(in_sc), d(in_sh), True, out=dw)
        outs.append((dy.cpu(), dw.cpu(), dgamma.cpu(), dbeta.cpu()))
    (dy_f, dw_f, dg_f, db_f), (dy_s, dw_s, dg_s, db_s) = outs
    close(dy_f, dy_ref, tol=1e-4)
    close(dy_f, dy_s, tol=2e-5)
    close(dw_f, dw_ref, tol=5e-4)
    close(dw_f, dw_s, tol=1e-4)
    assert torch.allclose(dg_f, dg_s, rtol=1e-5, atol=1e-5) and torch.allclose(db_f, db_s, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,cin,cout,h,w,k,acc", [(2, 32, 32, 64, 64, 3, False),
                                                  (2, 64, 32, 32, 32, 3, True),
                                                  (1, 256, 128, 28, 28, 3, False),   # masked 28-wide tiles
                                                  (3, 128, 128, 28, 28, 3, True),    # two-image strips, accumulate
                                                  (4, 64, 256, 28, 28, 3, False),    # strips, two cout tiles
                                                  (2, 16, 24, 20, 12, 3, True),      # ragged / scalar path
                                                  (2, 5, 7, 9, 11, 3, False)])
def test_conv_bn_backward_sums_epilogue(cuda, n, cin, cout, h, w, k, acc):
    """Input-gradient conv whose epilogue gathers the next BN backward's channel sums: same
    output as the plain conv, and dgamma / dbeta / dy equal to the two-pass BN backward."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(5 + n + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g).to(cuda)
    wt = (torch.randn(cin, k * k, cout, generator=g) / (cin * k * k) ** 0.5).to(cuda)
    y_bn = (torch.randn(n, cout, h, w, generator=g) * 1.3 + 0.2).to(cuda)
    base = torch.randn(n, cout, h, w, generator=g).to(cuda)
    gamma = (torch.rand(cout, generator=g) + 0.5).to(cuda)
    beta = (torch.randn(cout, generator=g) * 0.2).to(cuda)
    stats = torch.zeros(4, cout, device=cuda)
    nn.bn_train_stats(y_bn, gamma, beta, torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda),
                      stats, 0.99, 1e-3)
    out_a = base.clone() if acc else torch.empty_like(base)
    nn.conv2d(x, wt, k, out=out_a, accumulate=acc)
    out_b = base.clone() if acc else torch.empty_like(base)
    _, tsum = nn.conv2d_bnbwd(x, wt, k, y_bn, stats, True, out_b, accumulate=acc)
    assert torch.equal(out_a, out_b)
    res = []
    for ts in (None, tsum):
        dgamma, dbeta = torch.zeros(cout, device=cuda), torch.zeros(cout, device=cuda)
        dy = nn.bn_bwd(out_a, y_bn, stats, gamma, dgamma, dbeta, True, tile_sums=ts)
        res.append((dy.cpu(), dgamma.cpu(), dbeta.cpu()))
    scale = max(res[0][1].abs().max().item(), res[0][2].abs().max().item())
    assert (res[0][1] - res[1][1]).abs().max().item() <= 2e-5 * scale
    assert (res[0][2] - res[1][2]).abs().max().item() <= 2e-5 * scale
    close(res[1][0], res[0][0], tol=2e-5)


BF16_SHAPES = [  # n, cin, cout, h, w, k
    (2, 3, 32, 64, 64, 3),      # stem: 3 channels padded to one 16-channel chunk
    (2, 32, 32, 224, 224, 3),   # stage-1 geometry
    (2, 32, 64, 112, 112, 3),   # two output-channel blocks per workgroup; 3.5 tiles wide
    (2, 64, 128, 56, 56, 3),
    (3, 128, 256, 28, 28, 3),   # masked 28-wide tiles
    (2, 32, 64, 112, 112, 1),   # 1x1 projection
    (2, 128, 256, 28, 28, 1),
    (1, 16, 32, 12, 20, 3),     # partial tiles in both directions
    (2, 48, 64, 20, 24, 3),     # three 16-channel slices: the second staged 32-channel chunk is half empty
    (1, 40, 32, 9, 12, 3),      # channel count inside a slice: the padding comes back from the buffer bounds
    (2, 80, 64, 16, 16, 1),     # 1x1, five slices
    (1, 160, 96, 28, 28, 3),    # cout an odd multiple of 32 (32x16 tiles), five staged chunks
]


@pytest.mark.parametrize("n,cin,cout,h,w,k", BF16_SHAPES)
@pytest.mark.parametrize("prologue", [False, True])
def test_conv2d_bf16_forward(cuda, n, cin, cout, h, w, k, prologue):
    """bf16-operand / fp32-accumulate convolution (reduced-precision inference) against (a) the
    exact reference of its own arithmetic — torch conv2d in float64 on operands rounded to bf16 —
    to fp32 summation accuracy, and (b) the fp32 kernel within bf16 operand rounding."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cin, k * k, cout), generator=g) * (1.0 / (cin * k * k) ** 0.5)
    sc = sh = None
    if prologue:
        sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    xd, wd = x.to(cuda), wt.to(cuda)
    scd, shd = (sc.to(cuda), sh.to(cuda)) if prologue else (None, None)
    wp = nn.conv2d_bf16_weights(wd, k)
    got = nn.conv2d_bf16(xd, wp, cout, k, scd, shd, prologue).cpu()
    xe = x
    if prologue:   # the kernel's own prologue: fmaf then ReLU, in fp32
        xe = torch.relu(torch.addcmul(sh[None, :, None, None], x, sc[None, :, None, None]))
    xb = xe.to(torch.bfloat16).to(torch.float64)
    wb = iko_to_oihw(wt, k).to(torch.bfloat16).to(torch.float64)
    exact = F.conv2d(xb, wb, padding=k // 2).to(torch.float32)
    close(got, exact, tol=2e-5)
    fp32 = nn.conv2d(xd, wd, k, scd, shd, prologue).cpu()
    close(got, fp32, tol=2e-2)


def test_conv2d_bf16_rejects_unsupported_shapes(cuda):
    from leaffliction_amd import nn
    from leaffliction_amd._lib import LeafHipError
    wd = torch.zeros((8, 9, 32), device=cuda)
    wp = nn.conv2d_bf16_weights(wd, 3)
    with pytest.raises(LeafHipError):
        nn.conv2d_bf16(torch.zeros((1, 8, 6, 6), device=cuda), wp, 32, 3)        # width % 4 != 0
    wd2 = torch.zeros((8, 9, 16), device=cuda)
    with pytest.raises(LeafHipError):
        nn.conv2d_bf16(torch.zeros((1, 8, 8, 8), device=cuda), nn.conv2d_bf16_weights(wd2, 3), 16, 3)  # cout % 32


@pytest.mark.parametrize("xbf,ybf", [(True, True), (True, False), (False, True)])
def test_conv2d_bf16_activation_storage(cuda, xbf, ybf):
    """bf16 input and / or output storage of the reduced-precision convolution: reading a bf16
    tensor equals reading its fp32 widening bit for bit; a bf16 result is the fp32 result rounded
    to nearest even."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(3)
    n, cin, cout, h, w, k = 2, 64, 128, 56, 56, 3
    x = torch.randn((n, cin, h, w), generator=g).to(cuda)
    wt = (torch.randn((cin, k * k, cout), generator=g) * 0.05).to(cuda)
    sc, sh = (torch.rand(cin, generator=g) + 0.5).to(cuda), (torch.randn(cin, generator=g) * 0.3).to(cuda)
    wp = nn.conv2d_bf16_weights(wt, k)
    xin = x.to(torch.bfloat16) if xbf else x
    ref = nn.conv2d_bf16(xin.float() if xbf else x, wp, cout, k, sc, sh, True)          # fp32 in, fp32 out
    got = nn.conv2d_bf16(xin, wp, cout, k, sc, sh, True,
                         out_dtype=torch.bfloat16 if ybf else torch.float32)
    assert got.dtype == (torch.bfloat16 if ybf else torch.float32)
    assert torch.equal(got, ref.to(torch.bfloat16) if ybf else ref)


def test_conv2d_bf16_epilogue_and_passthrough_staging(cuda):
    """The folded-BatchNorm epilogue equals applying it to the fp32 result; a bf16 input without
    prologue (staged by interleaving the stored bits) equals the same values fed as fp32."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(8)
    for (n, cin, cout, h, w, k) in ((2, 64, 128, 56, 56, 3), (2, 32, 32, 40, 64, 3), (2, 128, 256, 28, 28, 1)):
        x = torch.randn((n, cin, h, w), generator=g).to(cuda).to(torch.bfloat16)
        wt = (torch.randn((cin, k * k, cout), generator=g) * 0.05).to(cuda)
        osc, osh = (torch.rand(cout, generator=g) + 0.5).to(cuda), (torch.randn(cout, generator=g) * 0.3).to(cuda)
        wp = nn.conv2d_bf16_weights(wt, k)
        raw = nn.conv2d_bf16(x.float(), wp, cout, k)                                   # fp32 in, fp32 out
        want = torch.relu(torch.addcmul(osh[None, :, None, None], raw, osc[None, :, None, None]))
        got = nn.conv2d_bf16(x, wp, cout, k, out_dtype=torch.bfloat16, out_scale=osc, out_shift=osh, out_relu=True)
        # the kernel's epilogue is one fmaf; torch's addcmul may round the product first: allow the
        # last fp32 bit, i.e. at most one bf16 step after rounding
        assert torch.allclose(got.float(), want, rtol=2.0 ** -7, atol=1e-6)
        assert (got.float() - want.to(torch.bfloat16).float()).abs().gt(0).float().mean().item() < 1e-3
        lin = nn.conv2d_bf16(x, wp, cout, k, out_scale=osc, out_shift=osh)              # no ReLU, fp32 out
        assert torch.allclose(lin, torch.addcmul(osh[None, :, None, None], raw, osc[None, :, None, None]),
                              rtol=1e-6, atol=1e-6)


def test_bf16_plane_kernels(cuda):
    """gap / residual-tail on bf16 tensors == the fp32 kernels on the widened tensors (then
    rounded for the bf16 output)."""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(4)
    n, c, h, w = 3, 64, 28, 28
    y = torch.randn((n, c, h, w), generator=g).to(cuda).to(torch.bfloat16)
    sc = torch.randn((n, c, h, w), generator=g).to(cuda).to(torch.bfloat16)
    a_s, a_b = (torch.rand(c, generator=g) + 0.5).to(cuda), (torch.randn(c, generator=g) * 0.2).to(cuda)
    k_s, k_b = (torch.rand(c, generator=g) + 0.5).to(cuda), (torch.randn(c, generator=g) * 0.2).to(cuda)
    gate = torch.rand((n, c), generator=g).to(cuda)
    m16 = nn.gap_bf16(y, a_s, a_b, True)
    m32 = nn.gap(y.float(), scale=a_s, shift=a_b, relu=True)
    assert (m16 - m32).abs().max().item() < 1e-5
    assert (nn.gap_bf16(y) - y.float().mean((2, 3))).abs().max().item() < 1e-5
    for sc_scale, sc_shift, sc_relu, s in ((k_s, k_b, True, gate), (k_s, k_b, False, gate), (None, None, False, None)):
        p16 = nn.block_tail_fwd_bf16(y, a_s, a_b, s, sc, sc_scale, sc_shift, sc_relu)
        route = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=cuda)
        p32 = torch.empty((n, c, h // 2, w // 2), device=cuda)
        nn.block_tail_fwd(y.float(), a_s, a_b, s, sc.float(), sc_scale, sc_shift, sc_relu, None, route, p32)
        assert torch.equal(p16, p32.to(torch.bfloat16))
    # already-activated input (a_scale = None): relu(shortcut + y * gate), pooled
    ya = torch.relu(y.float()).to(torch.bfloat16)
    p16 = nn.block_tail_fwd_bf16(ya, None, None, gate, sc, None, None, False)
    want = torch.nn.functional.max_pool2d(torch.relu(sc.float() + ya.float() * gate[:, :, None, None]), 2)
    assert torch.equal(p16, want.to(torch.bfloat16))


def test_conv2d_bf16_serves_as_input_gradient(cuda):
    """The bf16 forward kernel on the flipped / transposed weights (lf_conv2d_dgrad_weights_f32) is
    the reduced-precision input-gradient convolution: same result as the fp32 dgrad within bf16
    operand rounding.  (Groundwork for reduced-precision training; not used by the fp32 step.)"""
    from leaffliction_amd import nn
    g = torch.Generator().manual_seed(12)
    n, cin, cout, h, w, k = 2, 64, 128, 56, 56, 3
    dy = torch.randn((n, cout, h, w), generator=g).to(cuda)
    wt = (torch.randn((cin, k * k, cout), generator=g) * 0.05).to(cuda)
    wflip = nn.conv2d_dgrad_weights(wt, k)                       # [cout][k*k][cin]
    want = nn.conv2d(dy, wflip, k)
    got = nn.conv2d_bf16(dy, nn.conv2d_bf16_weights(wflip, k), cin, k)
    close(got.cpu(), want.cpu(), tol=2e-2)
    ref = F.conv_transpose2d(dy.cpu(), iko_to_oihw(wt.cpu(), k), padding=k // 2)
    close(got.cpu(), ref, tol=2e-2)
