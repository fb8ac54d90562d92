"""Host-side launchers for the leaf_cnn kernels of libleafhip.so (fp32, NCHW)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from .ops import _chk, _stream

_F32 = torch.float32
_ws_cache = {}
_ws_gen = 0
# A/B knob for measurements: route BatchNorm backward through the standalone apply kernel
_NO_FUSED_BN_WGRAD = os.environ.get("LEAFFLICTION_NO_FUSED_BN_WGRAD", "0") == "1"


def _workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    """Grow-only scratch buffer per device and slot (the C ABI never allocates).

    A buffer that is replaced goes back to torch's allocator, so every pointer taken from it is dead from then on.
    `workspace_generation()` counts the replacements: whoever keeps such pointers beyond the call — the HIP graph of
    the training step has them baked into its kernel nodes — compares generations and re-records (model/cnn.py)."""
    global _ws_gen
    key = (str(device), slot)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("a workspace would have to grow while a HIP graph is being recorded: run the shape "
                               "eagerly first (the warm-up steps size every workspace)")
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        _ws_gen += 1
    return buf


def workspace_generation() -> int:
    """Number of times a workspace buffer has been (re)allocated in this process; see _workspace."""
    return _ws_gen


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def conv2d(x: torch.Tensor, w_iko: torch.Tensor, ksize: int, in_scale=None, in_shift=None,
           in_relu: bool = False, out: Optional[torch.Tensor] = None,
           accumulate: bool = False) -> torch.Tensor:
    """y = conv2d_same(x', w); x' = relu?(x*in_scale[c]+in_shift[c]) if a prologue is given.

    x [N,Cin,H,W] f32, w_iko [Cin, k*k, Cout] f32 -> y [N,Cout,H,W].
    """
    _chk(x, _F32, "conv2d.x", 4)
    _chk(w_iko, _F32, "conv2d.w", 3)
    n, cin, h, w = x.shape
    if w_iko.shape[0] != cin or w_iko.shape[1] != ksize * ksize:
        raise ValueError(f"conv2d.w: expected [{cin},{ksize * ksize},Cout], got {tuple(w_iko.shape)}")
    cout = w_iko.shape[2]
    for t, nm in ((in_scale, "in_scale"), (in_shift, "in_shift")):
        if t is not None:
            _chk(t, _F32, f"conv2d.{nm}", 1)
            if t.shape[0] != cin:
                raise ValueError(f"conv2d.{nm}: expected [{cin}]")
    if out is None:
        if accumulate:
            raise ValueError("conv2d: accumulate needs an output tensor")
        out = torch.empty((n, cout, h, w), dtype=_F32, device=x.device)
    else:
        _chk(out, _F32, "conv2d.out", 4)
        if tuple(out.shape) != (n, cout, h, w):
            raise ValueError("conv2d.out: shape mismatch")
    _lib.call("lf_conv2d_f32", x.data_ptr(), w_iko.data_ptr(), out.data_ptr(), n, cin, h, w, cout,
              ksize, _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, 1 if accumulate else 0,
              _stream())
    return out


def conv2d_bf16_weights(w_iko: torch.Tensor, ksize: int) -> torch.Tensor:
    """Pack fp32 IKO weights [Cin, k*k, Cout] for conv2d_bf16 (uint16 view of bf16
    [ceil(Cin/16), k*k, Cout, 16])."""
    _chk(w_iko, _F32, "conv2d_bf16_weights.w", 3)
    cin, taps, cout = w_iko.shape
    if taps != ksize * ksize:
        raise ValueError(f"conv2d_bf16_weights.w: expected [{cin},{ksize * ksize},Cout]")
    n = int(_lib.load().lf_conv2d_bf16_weight_elems(cin, cout, ksize))
    out = torch.empty(n, dtype=torch.int16, device=w_iko.device)
    _lib.call("lf_conv2d_bf16_prep_weights", w_iko.data_ptr(), out.data_ptr(), cin, cout, ksize, _stream())
    return out


def conv2d_bf16(x: torch.Tensor, wprep: torch.Tensor, cout: int, ksize: int, in_scale=None, in_shift=None,
                in_relu: bool = False, out: Optional[torch.Tensor] = None,
                out_dtype: torch.dtype = torch.float32, out_scale=None, out_shift=None,
                out_relu: bool = False) -> torch.Tensor:
    """conv2d with bf16 operands / fp32 accumulation (inference).  x: fp32 or bf16 NCHW; the result
    is fp32 or bf16 NCHW (`out_dtype`, or the dtype of `out`); wprep from conv2d_bf16_weights.
    out_scale / out_shift [Cout] (+ out_relu): epilogue on the accumulators (folded BatchNorm)."""
    if x.dtype not in (_F32, torch.bfloat16):
        raise TypeError(f"conv2d_bf16.x: expected float32 or bfloat16, got {x.dtype}")
    _chk(x, x.dtype, "conv2d_bf16.x", 4)
    n, cin, h, w = x.shape
    if wprep.dtype != torch.int16 or wprep.numel() != ((cin + 15) // 16) * ksize * ksize * cout * 16:
        raise ValueError("conv2d_bf16.wprep: not the packed weights of this convolution")
    for t, nm in ((in_scale, "in_scale"), (in_shift, "in_shift")):
        if t is not None:
            _chk(t, _F32, f"conv2d_bf16.{nm}", 1)
            if t.shape[0] != cin:
                raise ValueError(f"conv2d_bf16.{nm}: expected [{cin}]")
    if out is None:
        if out_dtype not in (_F32, torch.bfloat16):
            raise TypeError("conv2d_bf16.out_dtype: float32 or bfloat16")
        out = torch.empty((n, cout, h, w), dtype=out_dtype, device=x.device)
    else:
        if out.dtype not in (_F32, torch.bfloat16):
            raise TypeError("conv2d_bf16.out: float32 or bfloat16")
        _chk(out, out.dtype, "conv2d_bf16.out", 4)
        if tuple(out.shape) != (n, cout, h, w):
            raise ValueError("conv2d_bf16.out: shape mismatch")
    _lib.call("lf_conv2d_bf16_act", x.data_ptr(), 1 if x.dtype == torch.bfloat16 else 0, wprep.data_ptr(),
              out.data_ptr(), 1 if out.dtype == torch.bfloat16 else 0, n, cin, h, w, cout, ksize,
              _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, _ptr(out_scale), _ptr(out_shift),
              1 if out_relu else 0, _stream())
    return out


def conv2d_bf16_mean(x: torch.Tensor, wprep: torch.Tensor, cout: int, ksize: int, out: torch.Tensor,
                     means: torch.Tensor, in_scale=None, in_shift=None, in_relu: bool = False, out_scale=None,
                     out_shift=None, out_relu: bool = False):
    """conv2d_bf16 with bf16 output AND the per-image channel means of the stored activation, taken in the
    convolution's epilogue (inference: a block's second convolution + the squeeze of its SE gate in one pass).
    x: fp32 or bf16 NCHW; out: bf16 [N,Cout,H,W]; means: fp32 [N,Cout].  Returns (out, means)."""
    if x.dtype not in (_F32, torch.bfloat16):
        raise TypeError(f"conv2d_bf16_mean.x: expected float32 or bfloat16, got {x.dtype}")
    _chk(x, x.dtype, "conv2d_bf16_mean.x", 4)
    n, cin, h, w = x.shape
    if wprep.dtype != torch.int16 or wprep.numel() != ((cin + 15) // 16) * ksize * ksize * cout * 16:
        raise ValueError("conv2d_bf16_mean.wprep: not the packed weights of this convolution")
    _chk(out, torch.bfloat16, "conv2d_bf16_mean.out", 4)
    _chk(means, _F32, "conv2d_bf16_mean.means", 2)
    if tuple(out.shape) != (n, cout, h, w) or tuple(means.shape) != (n, cout):
        raise ValueError("conv2d_bf16_mean: out [N,Cout,H,W] / means [N,Cout] shape mismatch")
    xb = 1 if x.dtype == torch.bfloat16 else 0
    ws = _workspace(int(_lib.load().lf_conv2d_bf16_act_mean_workspace(n, cin, h, w, cout, ksize, xb)), x.device, slot=1)
    _lib.call("lf_conv2d_bf16_act_mean", x.data_ptr(), xb, wprep.data_ptr(), out.data_ptr(), n, cin, h, w, cout, ksize,
              _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, _ptr(out_scale), _ptr(out_shift),
              1 if out_relu else 0, means.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return out, means


def gap_bf16(x: torch.Tensor, scale=None, shift=None, relu: bool = False,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N,C] fp32 plane means of relu?(x*scale[c]+shift[c]) for a bf16 NCHW tensor (inference)."""
    _chk(x, torch.bfloat16, "gap_bf16.x", 4)
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=_F32, device=x.device)
    _lib.call("lf_gap_bf16", x.data_ptr(), out.data_ptr(), n, c, h * w, _ptr(scale), _ptr(shift),
              1 if relu else 0, _stream())
    return out


def block_tail_fwd_bf16(y, a_scale, a_shift, s, sc, sc_scale, sc_shift, sc_relu: bool,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """maxpool2x2(relu(shortcut' + relu(y*a_scale+a_shift) * s)) on bf16 NCHW tensors (inference)."""
    _chk(y, torch.bfloat16, "block_tail_fwd_bf16.y", 4)
    _chk(sc, torch.bfloat16, "block_tail_fwd_bf16.sc", 4)
    n, c, h, w = y.shape
    if tuple(sc.shape) != (n, c, h, w):
        raise ValueError("block_tail_fwd_bf16.sc: shape mismatch")
    if out is None:
        out = torch.empty((n, c, h // 2, w // 2), dtype=torch.bfloat16, device=y.device)
    _lib.call("lf_block_tail_fwd_bf16", y.data_ptr(), _ptr(a_scale), _ptr(a_shift), _ptr(s),
              sc.data_ptr(), _ptr(sc_scale), _ptr(sc_shift), 1 if sc_relu else 0, out.data_ptr(), n, c, h, w,
              _stream())
    return out


# ---------------------------------------------------------------------------
# mixed-precision training step: bf16 storage, fp32 arithmetic (lf_*_bf16 / *_train_bf16)
# ---------------------------------------------------------------------------
_BF16 = torch.bfloat16


def conv2d_bf16_dgrad_weights(w_iko: torch.Tensor, ksize: int) -> torch.Tensor:
    """Packed bf16 weights of the input-gradient convolution of a conv with fp32 IKO weights w."""
    return conv2d_bf16_weights(conv2d_dgrad_weights(w_iko, ksize), ksize)


def conv2d_bf16_train(x: torch.Tensor, wprep: torch.Tensor, cout: int, ksize: int, out: torch.Tensor,
                      in_scale=None, in_shift=None, in_relu: bool = False, accumulate: bool = False,
                      stats: bool = False, pivot=None, mask_y=None, mask_scale=None, mask_shift=None,
                      mask_relu: bool = False):
    """Forward / input-gradient convolution of the bf16 training step: out (bf16 NCHW) = conv(x')
    (+ out when accumulate).  With stats=True returns (out, (tile_part, tiles)) — BatchNorm forward
    statistics about `pivot`, or (mask_y given) the backward sums of the BatchNorm out feeds."""
    if x.dtype not in (_F32, _BF16):
        raise TypeError(f"conv2d_bf16_train.x: expected float32 or bfloat16, got {x.dtype}")
    _chk(x, x.dtype, "conv2d_bf16_train.x", 4)
    _chk(out, _BF16, "conv2d_bf16_train.out", 4)
    n, cin, h, w = x.shape
    if wprep.dtype != torch.int16 or wprep.numel() != ((cin + 15) // 16) * ksize * ksize * cout * 16:
        raise ValueError("conv2d_bf16_train.wprep: not the packed weights of this convolution")
    if tuple(out.shape) != (n, cout, h, w):
        raise ValueError("conv2d_bf16_train.out: shape mismatch")
    for t, nm in ((in_scale, "in_scale"), (in_shift, "in_shift")):
        if t is not None:
            _chk(t, _F32, f"conv2d_bf16_train.{nm}", 1)
            if t.shape[0] != cin:
                raise ValueError(f"conv2d_bf16_train.{nm}: expected [{cin}]")
    tp, tiles = None, 0
    if mask_y is not None:
        _chk(mask_y, _BF16, "conv2d_bf16_train.mask_y", 4)
        if mask_y.shape != out.shape or mask_scale.shape[0] != cout or mask_shift.shape[0] != cout:
            raise ValueError("conv2d_bf16_train: mask shape mismatch")
        stats = True
    if stats:
        tiles = int(_lib.load().lf_conv2d_bf16_stats_tiles(n, cin, h, w, cout, ksize, 1 if x.dtype == _BF16 else 0))
        tp = _workspace(tiles * cout * 8, x.device, slot=1)
    _lib.call("lf_conv2d_bf16_train", x.data_ptr(), 1 if x.dtype == _BF16 else 0, wprep.data_ptr(),
              out.data_ptr(), n, cin, h, w, cout, ksize, _ptr(in_scale), _ptr(in_shift),
              1 if in_relu else 0, 1 if accumulate else 0, _ptr(tp), tp.numel() if tp is not None else 0,
              _ptr(pivot), _ptr(mask_y), _ptr(mask_scale), _ptr(mask_shift), 1 if mask_relu else 0,
              _stream())
    return (out, (tp, tiles)) if stats else out


def conv2d_bn_stats_bf16(x, wprep, cout: int, ksize: int, gamma, beta, mmean, mvar, stats: torch.Tensor,
                         in_scale=None, in_shift=None, in_relu: bool = False, out=None,
                         momentum: float = 0.99, eps: float = 1e-3) -> torch.Tensor:
    """Conv2D + training-mode BatchNormalization statistics on bf16 storage: the statistics are
    those of the ROUNDED output (what the next kernels read)."""
    n, _cin, h, w = x.shape
    out, (tp, tiles) = conv2d_bf16_train(x, wprep, cout, ksize, out, in_scale, in_shift, in_relu,
                                         stats=True, pivot=mmean)
    ws = _workspace(_lib.load().lf_bn_workspace(cout), x.device)
    _lib.call("lf_bn_train_stats_tiles_f32", tp.data_ptr(), tiles, n, cout, h * w, gamma.data_ptr(),
              beta.data_ptr(), mmean.data_ptr(), mvar.data_ptr(), float(momentum), float(eps),
              stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
              ws.data_ptr(), ws.numel(), _stream())
    return out


def bn_bwd_wgrad_bf16(x: torch.Tensor, g: torch.Tensor, y_bn: torch.Tensor, stats: torch.Tensor, gamma,
                      dgamma, dbeta, relu: bool, ksize: int, dw_out: torch.Tensor, dy_out,
                      in_scale=None, in_shift=None, in_relu: bool = False, alpha_nc=None, add_nc=None,
                      plane_g=None, plane_m=None, tile_sums=None):
    """bn_bwd_wgrad on bf16 tensors: the BatchNorm-backward sums come from per-plane sums
    (block_tail_bwd_bf16 / gap_stats_bf16) or per-tile sums (conv2d_bf16_train's epilogue) — never
    from another pass over g and y — and dY = BN'(g) is formed inside the weight-gradient kernel."""
    if x.dtype not in (_F32, _BF16):
        raise TypeError("bn_bwd_wgrad_bf16.x: float32 (stem) or bfloat16")
    _chk(x, x.dtype, "bn_bwd_wgrad_bf16.x", 4)
    _chk(g, _BF16, "bn_bwd_wgrad_bf16.g", 4)
    _chk(y_bn, _BF16, "bn_bwd_wgrad_bf16.y", 4)
    _chk(dw_out, _F32, "bn_bwd_wgrad_bf16.dw_out", 3)
    if dy_out is not None:
        _chk(dy_out, _BF16, "bn_bwd_wgrad_bf16.dy_out", 4)
    n, cin, h, w = x.shape
    cout = g.shape[1]
    if g.shape != y_bn.shape or (dy_out is not None and dy_out.shape != g.shape) or g.shape[0] != n \
            or tuple(g.shape[2:]) != (h, w) or tuple(dw_out.shape) != (cin, ksize * ksize, cout):
        raise ValueError("bn_bwd_wgrad_bf16: shape mismatch")
    if (tile_sums is None) == (plane_g is None):
        raise ValueError("bn_bwd_wgrad_bf16: exactly one of tile_sums / plane_g must be given")
    lib = _lib.load()
    coef = _workspace(5 * cout * 4, x.device, slot=2)
    ws = _workspace(lib.lf_bn_workspace(cout), x.device)
    if tile_sums is not None:
        if alpha_nc is not None or add_nc is not None:
            raise ValueError("bn_bwd_wgrad_bf16: tile_sums excludes alpha/add")
        tp, tiles = tile_sums
        _lib.call("lf_bn_bwd_sums_tiles_f32", tp.data_ptr(), tiles, stats[0].data_ptr(),
                  stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), gamma.data_ptr(),
                  dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), n, cout, h * w,
                  ws.data_ptr(), ws.numel(), _stream())
    else:
        _lib.call("lf_bn_bwd_sums_f32", None, _ptr(alpha_nc), _ptr(add_nc), None,
                  stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
                  1 if relu else 0, gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                  coef.data_ptr(), _ptr(plane_g), _ptr(plane_m), n, cout, h * w, ws.data_ptr(),
                  ws.numel(), _stream())
    ws = _workspace(lib.lf_conv2d_wgrad_bf16_workspace(n, cin, h, w, cout, ksize), x.device)
    _lib.call("lf_conv2d_wgrad_bf16", x.data_ptr(), g.data_ptr(), y_bn.data_ptr(), _ptr(alpha_nc),
              _ptr(add_nc), coef.data_ptr(), 1 if relu else 0, _ptr(dy_out), dw_out.data_ptr(), n, cin,
              h, w, cout, ksize, _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, ws.data_ptr(),
              ws.numel(), _stream())
    return dy_out


def conv2d_wgrad_bf16(x: torch.Tensor, dy: torch.Tensor, ksize: int, in_scale=None, in_shift=None,
                      in_relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw [Cin,k*k,Cout] (fp32) = sum x'[n,ci,y+ky-1,x+kx-1] * dy[n,co,y,x] from bf16 tensors."""
    _chk(dy, _BF16, "wgrad_bf16.dy", 4)
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    if out is None:
        out = torch.empty((cin, ksize * ksize, cout), dtype=_F32, device=x.device)
    ws = _workspace(_lib.load().lf_conv2d_wgrad_bf16_workspace(n, cin, h, w, cout, ksize), x.device)
    _lib.call("lf_conv2d_wgrad_bf16", x.data_ptr(), dy.data_ptr(), None, None, None, None, 0, None,
              out.data_ptr(), n, cin, h, w, cout, ksize, _ptr(in_scale), _ptr(in_shift),
              1 if in_relu else 0, ws.data_ptr(), ws.numel(), _stream())
    return out


def gap_stats_bf16(x: torch.Tensor, out=None, scale=None, shift=None, relu: bool = False, mask_sums=None):
    _chk(x, _BF16, "gap_stats_bf16.x", 4)
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=_F32, device=x.device)
    if mask_sums is not None and tuple(mask_sums.shape) != (n, c, 2):
        raise ValueError("gap_stats_bf16.mask_sums: expected [N,C,2]")
    _lib.call("lf_gap_stats_bf16", x.data_ptr(), out.data_ptr(), _ptr(mask_sums), n, c, h * w, _ptr(scale),
              _ptr(shift), 1 if relu else 0, _stream())
    return out


def block_tail_fwd_train_bf16(y, a_scale, a_shift, s, sc, sc_scale, sc_shift, sc_relu, drop, route, p):
    _chk(y, _BF16, "block_tail_fwd_train_bf16.y", 4)
    _chk(sc, _BF16, "block_tail_fwd_train_bf16.sc", 4)
    _chk(p, _BF16, "block_tail_fwd_train_bf16.p", 4)
    _chk(route, torch.uint8, "block_tail_fwd_train_bf16.route", 4)
    n, c, h, w = y.shape
    if sc.shape != y.shape or tuple(p.shape) != (n, c, h // 2, w // 2) or route.shape != p.shape:
        raise ValueError("block_tail_fwd_train_bf16: shape mismatch")
    _lib.call("lf_block_tail_fwd_train_bf16", y.data_ptr(), _ptr(a_scale), _ptr(a_shift), _ptr(s),
              sc.data_ptr(), _ptr(sc_scale), _ptr(sc_shift), 1 if sc_relu else 0, _ptr(drop),
              route.data_ptr(), p.data_ptr(), n, c, h, w, _stream())
    return route, p


def block_tail_bwd_bf16(dp, route, y, a_scale, a_shift, drop, dr, ds, plane_sums=None, sc_y=None,
                        sc_sums=None):
    _chk(dr, _BF16, "block_tail_bwd_bf16.dr", 4)
    _chk(dp, _BF16, "block_tail_bwd_bf16.dp", 4)
    _chk(route, torch.uint8, "block_tail_bwd_bf16.route", 4)
    n, c, h, w = dr.shape
    if tuple(dp.shape) != (n, c, h // 2, w // 2) or route.shape != dp.shape:
        raise ValueError("block_tail_bwd_bf16: shape mismatch")
    for t in (plane_sums, sc_sums):
        if t is not None and tuple(t.shape) != (n, c, 2):
            raise ValueError("block_tail_bwd_bf16 plane sums: expected [N,C,2]")
    _lib.call("lf_block_tail_bwd_bf16", dp.data_ptr(), route.data_ptr(), _ptr(y), _ptr(a_scale),
              _ptr(a_shift), _ptr(drop), dr.data_ptr(), _ptr(ds), _ptr(plane_sums), _ptr(sc_y),
              _ptr(sc_sums), n, c, h, w, _stream())
    return dr, ds


def bcast_planes_bf16(v, h, w, scale, out):
    _chk(v, _F32, "bcast_planes_bf16.v", 2)
    _chk(out, _BF16, "bcast_planes_bf16.out", 4)
    n, c = v.shape
    _lib.call("lf_bcast_planes_bf16", v.data_ptr(), out.data_ptr(), n * c, h * w, float(scale), _stream())
    return out


def cast_f32_bf16(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    _lib.call("lf_cast_f32_bf16", src.data_ptr(), dst.data_ptr(), src.numel(), _stream())
    return dst


def cast_bf16_f32(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    _lib.call("lf_cast_bf16_f32", src.data_ptr(), dst.data_ptr(), src.numel(), _stream())
    return dst


def conv2d_bn_stats(x: torch.Tensor, w_iko: torch.Tensor, ksize: int, gamma, beta, mmean, mvar,
                    stats: torch.Tensor, in_scale=None, in_shift=None, in_relu: bool = False,
                    out: Optional[torch.Tensor] = None, momentum: float = 0.99,
                    eps: float = 1e-3) -> torch.Tensor:
    """Conv2D + training-mode BatchNormalization statistics: y = conv(x'), and `stats` [4,C]
    (mean, invstd, scale, shift) + the moving statistics are produced from per-tile sums the
    convolution gathers in its epilogue, so y is not read again."""
    _chk(x, _F32, "conv2d_bn_stats.x", 4)
    _chk(w_iko, _F32, "conv2d_bn_stats.w", 3)
    n, cin, h, w = x.shape
    if w_iko.shape[0] != cin or w_iko.shape[1] != ksize * ksize:
        raise ValueError(f"conv2d_bn_stats.w: expected [{cin},{ksize * ksize},Cout]")
    cout = w_iko.shape[2]
    for t, nm in ((in_scale, "in_scale"), (in_shift, "in_shift")):
        if t is not None:
            _chk(t, _F32, f"conv2d_bn_stats.{nm}", 1)
            if t.shape[0] != cin:
                raise ValueError(f"conv2d_bn_stats.{nm}: expected [{cin}]")
    for t in (gamma, beta, mmean, mvar):
        _chk(t, _F32, "conv2d_bn_stats.param", 1)
        if t.shape[0] != cout:
            raise ValueError("conv2d_bn_stats: per-channel vectors must be [Cout]")
    _chk(stats, _F32, "conv2d_bn_stats.stats", 2)
    if tuple(stats.shape) != (4, cout):
        raise ValueError("conv2d_bn_stats.stats: expected [4,Cout]")
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=_F32, device=x.device)
    else:
        _chk(out, _F32, "conv2d_bn_stats.out", 4)
        if tuple(out.shape) != (n, cout, h, w):
            raise ValueError("conv2d_bn_stats.out: shape mismatch")
    lib = _lib.load()
    tiles = lib.lf_conv2d_stats_tiles(n, cin, h, w, cout, ksize)
    tp = _workspace(tiles * cout * 8, x.device, slot=1)
    _lib.call("lf_conv2d_stats_f32", x.data_ptr(), w_iko.data_ptr(), out.data_ptr(), n, cin, h, w,
              cout, ksize, _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, mmean.data_ptr(),
              tp.data_ptr(), tp.numel(), _stream())
    ws = _workspace(lib.lf_bn_workspace(cout), x.device)
    _lib.call("lf_bn_train_stats_tiles_f32", tp.data_ptr(), tiles, n, cout, h * w, gamma.data_ptr(),
              beta.data_ptr(), mmean.data_ptr(), mvar.data_ptr(), float(momentum), float(eps),
              stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
              ws.data_ptr(), ws.numel(), _stream())
    return out


def conv2d_bnbwd(x: torch.Tensor, w_iko: torch.Tensor, ksize: int, mask_y: torch.Tensor,
                 stats: torch.Tensor, relu: bool, out: torch.Tensor, accumulate: bool = False):
    """Input-gradient convolution out (+)= conv(x, w) whose result feeds the backward of the
    BatchNormalization with input mask_y / statistics `stats`: the epilogue also leaves the
    per-tile sums of that BN backward.  Returns (out, tile_sums) — pass tile_sums to
    bn_bwd_wgrad."""
    _chk(x, _F32, "conv2d_bnbwd.x", 4)
    _chk(w_iko, _F32, "conv2d_bnbwd.w", 3)
    _chk(mask_y, _F32, "conv2d_bnbwd.mask_y", 4)
    _chk(out, _F32, "conv2d_bnbwd.out", 4)
    n, cin, h, w = x.shape
    if w_iko.shape[0] != cin or w_iko.shape[1] != ksize * ksize:
        raise ValueError(f"conv2d_bnbwd.w: expected [{cin},{ksize * ksize},Cout]")
    cout = w_iko.shape[2]
    if tuple(out.shape) != (n, cout, h, w) or mask_y.shape != out.shape or tuple(stats.shape) != (4, cout):
        raise ValueError("conv2d_bnbwd: shape mismatch")
    tiles = _lib.load().lf_conv2d_stats_tiles(n, cin, h, w, cout, ksize)
    tp = _workspace(tiles * cout * 8, x.device, slot=1)
    _lib.call("lf_conv2d_bnbwd_f32", x.data_ptr(), w_iko.data_ptr(), out.data_ptr(), n, cin, h, w,
              cout, ksize, 1 if accumulate else 0, mask_y.data_ptr(), stats[2].data_ptr(),
              stats[3].data_ptr(), 1 if relu else 0, tp.data_ptr(), tp.numel(), _stream())
    return out, (tp, tiles)


def conv2d_dgrad_weights(w_iko: torch.Tensor, ksize: int) -> torch.Tensor:
    """[Cin,k*k,Cout] -> [Cout,k*k(flipped),Cin]: conv2d(dy, wt) is the input gradient."""
    _chk(w_iko, _F32, "dgrad_weights.w", 3)
    cin, taps, cout = w_iko.shape
    if taps != ksize * ksize:
        raise ValueError("dgrad_weights: taps != ksize^2")
    wt = torch.empty((cout, taps, cin), dtype=_F32, device=w_iko.device)
    _lib.call("lf_conv2d_dgrad_weights_f32", w_iko.data_ptr(), wt.data_ptr(), cin, ksize, cout,
              _stream())
    return wt


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, ksize: int, in_scale=None, in_shift=None,
                 in_relu: bool = False, out: Optional[torch.Tensor] = None,
                 beta: float = 0.0) -> torch.Tensor:
    """dw [Cin,k*k,Cout] = sum_{n,y,x} x'[n,ci,y+ky-1,x+kx-1] * dy[n,co,y,x] (+ beta*out)."""
    _chk(x, _F32, "wgrad.x", 4)
    _chk(dy, _F32, "wgrad.dy", 4)
    n, cin, h, w = x.shape
    if dy.shape[0] != n or tuple(dy.shape[2:]) != (h, w):
        raise ValueError("wgrad: x and dy must share N,H,W")
    cout = dy.shape[1]
    if out is None:
        out = torch.empty((cin, ksize * ksize, cout), dtype=_F32, device=x.device)
        beta = 0.0
    else:
        _chk(out, _F32, "wgrad.out", 3)
        if tuple(out.shape) != (cin, ksize * ksize, cout):
            raise ValueError("wgrad.out: shape mismatch")
    nbytes = _lib.load().lf_conv2d_wgrad_workspace(n, cin, h, w, cout, ksize)
    ws = _workspace(nbytes, x.device)
    _lib.call("lf_conv2d_wgrad_f32", x.data_ptr(), dy.data_ptr(), n, cin, h, w, cout, ksize,
              _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, ws.data_ptr(), ws.numel(),
              _stream())
    _lib.call("lf_conv2d_wgrad_reduce_f32", ws.data_ptr(), out.data_ptr(), n, cin, h, w, cout,
              ksize, float(beta), _stream())
    return out


def bn_bwd_wgrad(x: torch.Tensor, g: torch.Tensor, y_bn: torch.Tensor, stats: torch.Tensor, gamma,
                 dgamma, dbeta, relu: bool, ksize: int, dw_out: torch.Tensor, dy_out: torch.Tensor,
                 in_scale=None, in_shift=None, in_relu: bool = False, alpha_nc=None, add_nc=None,
                 plane_g=None, plane_m=None, tile_sums=None) -> torch.Tensor:
    """BatchNormalization backward of g (BN input y_bn) followed by the weight gradient of the
    convolution that produced y_bn from x:  dy = BN'(g) is formed inside the wgrad kernel from
    g and y_bn (and written to dy_out, when given, for the input-gradient convolution); dgamma / dbeta /
    dw_out are filled.  Falls back to the two-kernel route for shapes the fused kernel does not
    take (same results up to rounding)."""
    _chk(x, _F32, "bn_bwd_wgrad.x", 4)
    _chk(g, _F32, "bn_bwd_wgrad.g", 4)
    _chk(y_bn, _F32, "bn_bwd_wgrad.y", 4)
    if dy_out is not None:
        _chk(dy_out, _F32, "bn_bwd_wgrad.dy_out", 4)
    _chk(dw_out, _F32, "bn_bwd_wgrad.dw_out", 3)
    n, cin, h, w = x.shape
    cout = g.shape[1]
    if g.shape != y_bn.shape or (dy_out is not None and dy_out.shape != g.shape) or g.shape[0] != n \
            or tuple(g.shape[2:]) != (h, w):
        raise ValueError("bn_bwd_wgrad: shape mismatch")
    if tuple(dw_out.shape) != (cin, ksize * ksize, cout):
        raise ValueError("bn_bwd_wgrad.dw_out: shape mismatch")
    lib = _lib.load()
    if _NO_FUSED_BN_WGRAD or not lib.lf_conv2d_wgrad_bn_supported(n, cin, h, w, cout, ksize):
        dy = bn_bwd(g, y_bn, stats, gamma, dgamma, dbeta, relu, alpha_nc=alpha_nc, add_nc=add_nc,
                    out=dy_out, plane_g=plane_g, plane_m=plane_m, tile_sums=tile_sums)
        conv2d_wgrad(x, dy, ksize, in_scale, in_shift, in_relu, out=dw_out)
        return dy_out
    for t in (plane_g, plane_m):
        if t is not None and tuple(t.shape) != (n, cout, 2):
            raise ValueError("bn_bwd_wgrad: plane sums must be [N,C,2]")
    for t in (alpha_nc, add_nc):
        if t is not None and tuple(t.shape) != (n, cout):
            raise ValueError("bn_bwd_wgrad: alpha/add must be [N,C]")
    coef = _workspace(5 * cout * 4, x.device, slot=2)
    ws = _workspace(lib.lf_bn_workspace(cout), x.device)
    if tile_sums is not None:
        if alpha_nc is not None or add_nc is not None or plane_g is not None:
            raise ValueError("bn_bwd_wgrad: tile_sums excludes alpha/add/plane sums")
        tp, tiles = tile_sums
        _lib.call("lf_bn_bwd_sums_tiles_f32", tp.data_ptr(), tiles, stats[0].data_ptr(),
                  stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), gamma.data_ptr(),
                  dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), n, cout, h * w,
                  ws.data_ptr(), ws.numel(), _stream())
    else:
        _lib.call("lf_bn_bwd_sums_f32", g.data_ptr(), _ptr(alpha_nc), _ptr(add_nc), y_bn.data_ptr(),
                  stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
                  1 if relu else 0, gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                  coef.data_ptr(), _ptr(plane_g), _ptr(plane_m), n, cout, h * w, ws.data_ptr(),
                  ws.numel(), _stream())
    ws = _workspace(lib.lf_conv2d_wgrad_workspace(n, cin, h, w, cout, ksize), x.device)
    _lib.call("lf_conv2d_wgrad_bn_f32", x.data_ptr(), g.data_ptr(), y_bn.data_ptr(), _ptr(alpha_nc),
              _ptr(add_nc), coef.data_ptr(), 1 if relu else 0, _ptr(dy_out), n, cin, h, w, cout,
              ksize, _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, ws.data_ptr(), ws.numel(),
              _stream())
    _lib.call("lf_conv2d_wgrad_reduce_f32", ws.data_ptr(), dw_out.data_ptr(), n, cin, h, w, cout,
              ksize, 0.0, _stream())
    return dy_out


# ---------------------------------------------------------------------------
# non-conv layers
# ---------------------------------------------------------------------------
def input_stage(x_u8: torch.Tensor, aug4: torch.Tensor, mean=None, denom=None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """u8 [N,H,W,3] -> f32 [N,3,H,W] with flip/rotate/contrast (aug4 [N,4]) + Normalization."""
    _chk(x_u8, torch.uint8, "input_stage.x", 4)
    _chk(aug4, _F32, "input_stage.aug", 2)
    n, h, w, c = x_u8.shape
    if c != 3 or tuple(aug4.shape) != (n, 4):
        raise ValueError("input_stage: x must be [N,H,W,3] and aug [N,4]")
    if out is None:
        out = torch.empty((n, 3, h, w), dtype=_F32, device=x_u8.device)
    ws = torch.empty((n, 24), dtype=_F32, device=x_u8.device)
    m = d = None
    if mean is not None:
        m = (_lib.c_float * 3)(*[float(v) for v in mean])
        d = (_lib.c_float * 3)(*[float(v) for v in denom])
    _lib.call("lf_input_stage_f32", x_u8.data_ptr(), out.data_ptr(), n, h, w, aug4.data_ptr(), m, d,
              ws.data_ptr(), _stream())
    return out


def scale_shift_act(x, scale, shift, relu: bool, out=None):
    _chk(x, _F32, "scale_shift_act.x", 4)
    n, c, h, w = x.shape
    for t in (scale, shift):
        _chk(t, _F32, "scale_shift_act.scale/shift", 1)
        if t.shape[0] != c:
            raise ValueError("scale_shift_act: per-channel vectors must be [C]")
    if out is None:
        out = torch.empty_like(x)
    _lib.call("lf_scale_shift_act_f32", x.data_ptr(), out.data_ptr(), n, c, h * w, scale.data_ptr(),
              shift.data_ptr(), 1 if relu else 0, _stream())
    return out


def bn_train_stats(y, gamma, beta, mmean, mvar, stats, momentum=0.99, eps=1e-3):
    """stats: f32 [4,C] rows = mean, invstd, scale, shift (written)."""
    _chk(y, _F32, "bn_train_stats.y", 4)
    n, c, h, w = y.shape
    for t in (gamma, beta, mmean, mvar):
        _chk(t, _F32, "bn_train_stats.param", 1)
        if t.shape[0] != c:
            raise ValueError("bn_train_stats: per-channel vectors must be [C]")
    _chk(stats, _F32, "bn_train_stats.stats", 2)
    if tuple(stats.shape) != (4, c):
        raise ValueError("bn_train_stats.stats: expected [4,C]")
    nbytes = _lib.load().lf_bn_workspace(c)
    ws = _workspace(nbytes, y.device)
    _lib.call("lf_bn_train_stats_f32", y.data_ptr(), n, c, h * w, gamma.data_ptr(), beta.data_ptr(),
              mmean.data_ptr(), mvar.data_ptr(), float(momentum), float(eps), stats[0].data_ptr(),
              stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), ws.data_ptr(),
              ws.numel(), _stream())
    return stats


def bn_infer_scale_shift(gamma, beta, mmean, mvar, stats, eps=1e-3):
    c = gamma.shape[0]
    _lib.call("lf_bn_infer_scale_shift_f32", c, gamma.data_ptr(), beta.data_ptr(), mmean.data_ptr(),
              mvar.data_ptr(), float(eps), stats[2].data_ptr(), stats[3].data_ptr(), _stream())
    return stats


def bn_bwd(g, y, stats, gamma, dgamma, dbeta, relu: bool, alpha_nc=None, add_nc=None, out=None,
           plane_g=None, plane_m=None, tile_sums=None):
    """BatchNorm backward; the ReLU mask (relu=True) is recomputed from y and stats[2:4].
    plane_g ([N,C,2] from block_tail_bwd) / plane_m ([N,C,2] from gap) replace the reduction
    pass over g and y; so do tile_sums (from conv2d_bnbwd, which produced g)."""
    _chk(g, _F32, "bn_bwd.g", 4)
    _chk(y, _F32, "bn_bwd.y", 4)
    if g.shape != y.shape:
        raise ValueError("bn_bwd: g and y must share a shape")
    n, c, h, w = y.shape
    for t in (plane_g, plane_m):
        if t is not None:
            _chk(t, _F32, "bn_bwd.plane sums", 3)
            if tuple(t.shape) != (n, c, 2):
                raise ValueError("bn_bwd: plane sums must be [N,C,2]")
    for t in (alpha_nc, add_nc):
        if t is not None:
            _chk(t, _F32, "bn_bwd.alpha/add", 2)
            if tuple(t.shape) != (n, c):
                raise ValueError("bn_bwd: alpha/add must be [N,C]")
    if out is None:
        out = torch.empty_like(y)
    ws = _workspace(_lib.load().lf_bn_workspace(c), y.device)
    have = 0
    if tile_sums is not None:
        if alpha_nc is not None or add_nc is not None or plane_g is not None:
            raise ValueError("bn_bwd: tile_sums excludes alpha/add/plane sums")
        tp, tiles = tile_sums
        coef = _workspace(5 * c * 4, y.device, slot=2)
        _lib.call("lf_bn_bwd_sums_tiles_f32", tp.data_ptr(), tiles, stats[0].data_ptr(),
                  stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), gamma.data_ptr(),
                  dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), n, c, h * w, ws.data_ptr(),
                  ws.numel(), _stream())
        have = 1
    _lib.call("lf_bn_bwd_f32", g.data_ptr(), _ptr(alpha_nc), _ptr(add_nc), y.data_ptr(),
              stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
              1 if relu else 0, gamma.data_ptr(), out.data_ptr(), dgamma.data_ptr(),
              dbeta.data_ptr(), _ptr(plane_g), _ptr(plane_m), have, n, c, h * w, ws.data_ptr(),
              ws.numel(), _stream())
    return out


def gap(x, out=None, scale=None, shift=None, relu: bool = False, mask_sums=None):
    """[N,C,H,W] -> [N,C] mean of act(x*scale[c]+shift[c]) (plain mean without scale).
    mask_sums [N,C,2] (optional) receives {count of x*scale+shift > 0, sum of x over those}."""
    _chk(x, _F32, "gap.x", 4)
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=_F32, device=x.device)
    if mask_sums is not None:
        _chk(mask_sums, _F32, "gap.mask_sums", 3)
        if tuple(mask_sums.shape) != (n, c, 2):
            raise ValueError("gap.mask_sums: expected [N,C,2]")
    _lib.call("lf_gap_f32", x.data_ptr(), out.data_ptr(), n * c, h * w, c, _ptr(scale), _ptr(shift),
              1 if relu else 0, _ptr(mask_sums), _stream())
    return out


def bcast_planes(v, h, w, scale, out=None):
    _chk(v, _F32, "bcast_planes.v", 2)
    n, c = v.shape
    if out is None:
        out = torch.empty((n, c, h, w), dtype=_F32, device=v.device)
    _lib.call("lf_bcast_planes_f32", v.data_ptr(), out.data_ptr(), n * c, h * w, float(scale),
              _stream())
    return out


def se_fwd(m, w1, b1, w2, b2, z1, s):
    n, c = m.shape
    cr = w1.shape[1]
    if tuple(w1.shape) != (c, cr) or tuple(w2.shape) != (cr, c) or tuple(z1.shape) != (n, cr) \
            or tuple(s.shape) != (n, c):
        raise ValueError("se_fwd: shape mismatch")
    _lib.call("lf_se_fwd_f32", m.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
              b2.data_ptr(), z1.data_ptr(), s.data_ptr(), n, c, cr, _stream())
    return s


def se_bwd(ds, m, z1, s, w1, w2, dm, dw1, db1, dw2, db2, dm_scale: float = 1.0):
    n, c = m.shape
    cr = w1.shape[1]
    ws = _workspace(_lib.load().lf_se_bwd_workspace(n, c, cr), m.device)
    _lib.call("lf_se_bwd_f32", ds.data_ptr(), m.data_ptr(), z1.data_ptr(), s.data_ptr(),
              w1.data_ptr(), w2.data_ptr(), dm.data_ptr(), dw1.data_ptr(), db1.data_ptr(),
              dw2.data_ptr(), db2.data_ptr(), n, c, cr, float(dm_scale), ws.data_ptr(), ws.numel(),
              _stream())
    return dm


def block_tail_fwd(y, a_scale, a_shift, s, sc, sc_scale, sc_shift, sc_relu, drop, route, p):
    """Add -> ReLU -> SpatialDropout2D -> MaxPool2D(2); route: uint8 [N,C,H/2,W/2] (written)."""
    _chk(y, _F32, "block_tail_fwd.y", 4)
    _chk(route, torch.uint8, "block_tail_fwd.route", 4)
    n, c, h, w = y.shape
    if sc.shape != y.shape or tuple(p.shape) != (n, c, h // 2, w // 2) or route.shape != p.shape:
        raise ValueError("block_tail_fwd: shape mismatch")
    _lib.call("lf_block_tail_fwd_f32", y.data_ptr(), _ptr(a_scale), _ptr(a_shift), _ptr(s),
              sc.data_ptr(), _ptr(sc_scale), _ptr(sc_shift), 1 if sc_relu else 0, _ptr(drop),
              route.data_ptr(), p.data_ptr(), n, c, h, w, _stream())
    return route, p


def block_tail_bwd(dp, route, y, a_scale, a_shift, drop, dr, ds, plane_sums=None, sc_y=None,
                   sc_sums=None):
    _chk(dr, _F32, "block_tail_bwd.dr", 4)
    _chk(route, torch.uint8, "block_tail_bwd.route", 4)
    n, c, h, w = dr.shape
    if tuple(dp.shape) != (n, c, h // 2, w // 2) or route.shape != dp.shape:
        raise ValueError("block_tail_bwd: shape mismatch")
    for t in (plane_sums, sc_sums):
        if t is not None and tuple(t.shape) != (n, c, 2):
            raise ValueError("block_tail_bwd plane sums: expected [N,C,2]")
    if sc_y is not None and sc_y.shape != dr.shape:
        raise ValueError("block_tail_bwd.sc_y: shape mismatch")
    _lib.call("lf_block_tail_bwd_f32", dp.data_ptr(), route.data_ptr(), _ptr(y), _ptr(a_scale),
              _ptr(a_shift), _ptr(drop), dr.data_ptr(), _ptr(ds), _ptr(plane_sums), _ptr(sc_y),
              _ptr(sc_sums), n, c, h, w, _stream())
    return dr, ds


def head_fwd(feat, w, b, ytrue, probs, loss):
    n, f = feat.shape
    c = w.shape[1]
    _lib.call("lf_head_fwd_f32", feat.data_ptr(), w.data_ptr(), b.data_ptr(), _ptr(ytrue),
              probs.data_ptr(), _ptr(loss), n, f, c, _stream())
    return probs


def head_bwd(feat, w, probs, ytrue, dlogits, dfeat, dw, db, inv_n):
    n, f = feat.shape
    c = w.shape[1]
    _lib.call("lf_head_bwd_f32", feat.data_ptr(), w.data_ptr(), probs.data_ptr(), ytrue.data_ptr(),
              dlogits.data_ptr(), dfeat.data_ptr(), dw.data_ptr(), db.data_ptr(), n, f, c,
              float(inv_n), _stream())


def mul(a, b, out):
    if a.shape != b.shape or out.shape != a.shape:
        raise ValueError("mul: shape mismatch")
    _lib.call("lf_mul_f32", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
    return out


def adamw_step(param, grad, m, v, ema, offsets, l2, max_count, lr, step, beta1=0.9, beta2=0.999,
               eps=1e-7, weight_decay=1e-4, clipnorm=0.5, ema_decay=0.999, ema_copy=False,
               norms=None):
    nt = offsets.numel() - 1
    if norms is None:
        norms = torch.empty(nt, dtype=_F32, device=param.device)
    ws = _workspace(_lib.load().lf_adamw_workspace(nt), param.device)
    _lib.call("lf_adamw_step_f32", param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(),
              _ptr(ema), offsets.data_ptr(), l2.data_ptr(), nt, int(max_count), float(lr),
              float(beta1), float(beta2), float(eps), float(weight_decay), float(clipnorm),
              int(step), float(ema_decay), 1 if ema_copy else 0, norms.data_ptr(), ws.data_ptr(),
              ws.numel(), _stream())
    return norms


def ema_update(ema, w, decay, copy):
    _lib.call("lf_ema_update_f32", ema.data_ptr(), w.data_ptr(), w.numel(), float(decay),
              1 if copy else 0, _stream())
