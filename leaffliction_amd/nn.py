"""Host-side launchers for the leaf_cnn kernels of libleafhip.so (fp32, NCHW)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .ops import _chk, _stream

_F32 = torch.float32
_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per device (the C ABI never allocates)."""
    key = str(device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def conv2d(x: torch.Tensor, w_iko: torch.Tensor, ksize: int, in_scale=None, in_shift=None,
           in_relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
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
        out = torch.empty((n, cout, h, w), dtype=_F32, device=x.device)
    else:
        _chk(out, _F32, "conv2d.out", 4)
        if tuple(out.shape) != (n, cout, h, w):
            raise ValueError("conv2d.out: shape mismatch")
    _lib.call("lf_conv2d_f32", x.data_ptr(), w_iko.data_ptr(), out.data_ptr(), n, cin, h, w, cout,
              ksize, _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, _stream())
    return out


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
    _lib.call("lf_conv2d_wgrad_f32", x.data_ptr(), dy.data_ptr(), out.data_ptr(), n, cin, h, w,
              cout, ksize, _ptr(in_scale), _ptr(in_shift), 1 if in_relu else 0, float(beta),
              ws.data_ptr(), ws.numel(), _stream())
    return out
