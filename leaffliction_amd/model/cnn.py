"""leaf_cnn on MI355X: the reference's residual-SE CNN (srcs/model/cnn.py:9-131) with a
hand-written forward/backward over libleafhip's kernels.

`build_leafcnn(...)` keeps the reference's signature and returns `(model, norm_layer)`;
the model object offers what the reference's callers use of a Keras model
(SURVEY §8b "Model object contract"): `fit`, `evaluate`, `predict`, `get_weights`,
`set_weights`, `save`, `stop_training` — plus `train_step` for the benchmark.

Data layout in HBM: activations NCHW f32; every trainable tensor lives in ONE flat f32
buffer (`flat_p`, with matching `flat_g`, Adam `flat_m`/`flat_v` and `flat_ema`), so the
optimizer is two launches and data-parallel training all-reduces one bucket.  Conv kernels
are stored "IKO" [Cin, k*k, Cout]; `get_weights()` converts to keras HWIO.
"""
from __future__ import annotations

import json
import os
import math
import zipfile
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import nn, ops

BN_MOMENTUM = 0.99
BN_EPS = 1e-3
NORM_EPS = 1e-7


def _specs(num_classes: int, widths: List[int], use_se: bool):
    """Trainable tensors in creation order: (name, shape, kind).  kind 'w3' = 3x3 kernel
    (carries the L2 kernel_regularizer, cnn.py:21-29), 'w1' = 1x1 kernel, 'vec', 'dense'."""
    out = [("stem.w", (3, 9, widths[0]), "w3"), ("stem.bn.gamma", (widths[0],), "vec"),
           ("stem.bn.beta", (widths[0],), "vec")]
    cin = widths[0]
    for i, f in enumerate(widths):
        p = f"s{i}."
        out += [(p + "c1.w", (cin, 9, f), "w3"), (p + "bn1.gamma", (f,), "vec"),
                (p + "bn1.beta", (f,), "vec"), (p + "c2.w", (f, 9, f), "w3"),
                (p + "bn2.gamma", (f,), "vec"), (p + "bn2.beta", (f,), "vec")]
        if use_se:
            out += [(p + "se.w1", (f, f // 8), "w1"), (p + "se.b1", (f // 8,), "vec"),
                    (p + "se.w2", (f // 8, f), "w1"), (p + "se.b2", (f,), "vec")]
        if cin != f:
            out += [(p + "proj.w", (cin, 1, f), "w1"), (p + "bnp.gamma", (f,), "vec"),
                    (p + "bnp.beta", (f,), "vec")]
        cin = f
    out += [("dense.w", (widths[-1], num_classes), "dense"), ("dense.b", (num_classes,), "vec")]
    return out


def _bn_layers(widths: List[int]):
    out = [("stem.bn", widths[0])]
    cin = widths[0]
    for i, f in enumerate(widths):
        out += [(f"s{i}.bn1", f), (f"s{i}.bn2", f)]
        if cin != f:
            out.append((f"s{i}.bnp", f))
        cin = f
    return out


class Normalization:
    """keras.layers.Normalization(axis=-1) stand-in: per-channel mean/variance, adapt()."""

    def __init__(self) -> None:
        self.mean = np.zeros(3, np.float32)
        self.variance = np.ones(3, np.float32)
        self.adapted = False

    def adapt(self, data: np.ndarray) -> None:
        """data: [N,H,W,3] float32 in [0,1] (the loader's output)."""
        d = np.asarray(data, dtype=np.float64).reshape(-1, data.shape[-1])
        self.mean = d.mean(axis=0).astype(np.float32)
        self.variance = d.var(axis=0).astype(np.float32)
        self.adapted = True

    @property
    def denom(self) -> np.ndarray:
        return np.maximum(np.sqrt(self.variance), NORM_EPS).astype(np.float32)


class LeafCNN:
    name = "leaf_cnn"
    _mut = 0                            # see __init__ (class-level defaults: subclasses that skip it still step)
    _infer_cache: Dict[str, Any] = {}

    def __init__(self, *, num_classes: int, img_size: int = 224, use_norm: bool = True,
                 widths: Optional[List[int]] = None, drop_block: float = 0.15,
                 drop_top: float = 0.40, l2_reg: float = 0.0, separable: bool = False,
                 augment: bool = True, use_se: bool = True, seed: int = 0,
                 device: Optional[torch.device] = None) -> None:
        if separable:
            # cnn.py:22-25 passes `kernel_regularizer=` to keras.layers.SeparableConv2D, which
            # Keras 3 (requirements: keras>=3) rejects as an unrecognised keyword with a
            # ValueError — `train --separable` therefore logs the error and returns in the
            # reference too (train.py:471-473).  Same outcome here, no depthwise kernels.
            raise ValueError("--separable: SeparableConv2D is not available (the reference's "
                             "own call is rejected by Keras 3; no depthwise path is provided)")
        if not torch.cuda.is_available():
            raise RuntimeError("LeafCNN needs a HIP device: there is no CPU fallback")
        self.num_classes = int(num_classes)
        self.img_size = int(img_size)
        self.widths = list(widths or [32, 64, 128])
        self.drop_block = float(drop_block or 0.0)
        self.drop_top = float(drop_top or 0.0)
        self.l2_reg = float(l2_reg or 0.0)
        self.augment = bool(augment)
        self.use_se = bool(use_se)
        self.infer_dtype = os.environ.get("LEAFFLICTION_INFER_DTYPE", "f32")  # see set_inference_dtype
        if self.infer_dtype not in ("f32", "bf16"):
            raise ValueError("LEAFFLICTION_INFER_DTYPE must be f32 or bf16")
        self.train_dtype = os.environ.get("LEAFFLICTION_TRAIN_DTYPE", "f32")  # see set_training_dtype
        if self.train_dtype not in ("f32", "bf16"):
            raise ValueError("LEAFFLICTION_TRAIN_DTYPE must be f32 or bf16")
        # (the data-parallel gradient bucket's dtype is train.parallel.DataParallel.bucket_dtype)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.norm = Normalization() if use_norm else None
        self.stop_training = False
        self._mut = 0            # passes that wrote parameters / moving statistics from inside the library
        self._infer_cache: Dict[str, Any] = {}
        self.gen = torch.Generator(device="cpu").manual_seed(int(seed))
        self.np_rng = np.random.RandomState(int(seed) & 0x7FFFFFFF)
        from ..utils.system_info import cap_torch_threads
        cap_torch_threads()   # the per-step host draws are small CPU tensor ops: see there

        # ---- flat parameter / state storage
        self.specs = _specs(self.num_classes, self.widths, self.use_se)
        offs, off = [0], 0
        for _n, shape, _k in self.specs:
            off += int(np.prod(shape))
            offs.append(off)
        self.n_params = off
        dev = self.device
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.flat_ema = torch.zeros_like(self.flat_p)
        self.offsets = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.l2_vec = torch.tensor([self.l2_reg if k == "w3" else 0.0 for _n, _s, k in self.specs],
                                   dtype=torch.float32, device=dev)
        self.max_count = max(int(np.prod(s)) for _n, s, _k in self.specs)
        self.norms_ws = torch.empty(len(self.specs), dtype=torch.float32, device=dev)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for (name, shape, _k), b, e in zip(self.specs, offs[:-1], offs[1:]):
            self.p[name] = self.flat_p[b:e].view(shape)
            self.g[name] = self.flat_g[b:e].view(shape)
        self.bn_layers = _bn_layers(self.widths)
        soff = 0
        for _n, c in self.bn_layers:
            soff += 2 * c
        self.flat_s = torch.zeros(soff, dtype=torch.float32, device=dev)
        self.flat_s_ema = torch.zeros_like(self.flat_s)
        self.s: Dict[str, torch.Tensor] = {}
        self.stats: Dict[str, torch.Tensor] = {}
        soff = 0
        for n_, c in self.bn_layers:
            self.s[n_ + ".mean"] = self.flat_s[soff:soff + c]
            self.s[n_ + ".var"] = self.flat_s[soff + c:soff + 2 * c]
            soff += 2 * c
            self.stats[n_] = torch.zeros((4, c), dtype=torch.float32, device=dev)
        self._init_weights()
        self.opt_step = 0
        self.ema_started = False
        self._bufs: Dict[Any, Dict[str, torch.Tensor]] = {}
        self._stage: Dict[Any, Dict[str, Any]] = {}
        self._wt: Dict[str, torch.Tensor] = {}
        self._saved: Dict[str, Any] = {}
        self._global_n: Optional[int] = None
        self._compiled: Dict[str, Any] = {}
        self._graphs: Dict[Any, Dict[str, Any]] = {}
        self._graphs_on = os.environ.get("LEAFFLICTION_GRAPH", "1") != "0"

    # ------------------------------------------------------------------ init
    def _init_weights(self) -> None:
        """glorot_uniform kernels, zero biases, BN gamma=1/beta=0, moving mean 0 / var 1."""
        for name, shape, kind in self.specs:
            if kind == "vec":
                self.p[name].fill_(1.0 if name.endswith("gamma") else 0.0)
                continue
            if len(shape) == 3:
                fan_in, fan_out = shape[0] * shape[1], shape[2] * shape[1]
            else:
                fan_in, fan_out = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = (torch.rand(shape, generator=self.gen) * 2 - 1) * lim
            self.p[name].copy_(w)
        for n_, _c in self.bn_layers:
            self.s[n_ + ".mean"].zero_()
            self.s[n_ + ".var"].fill_(1.0)

    # ------------------------------------------------------------- buffers
    def _buf(self, n: int, key: str, shape, dtype=torch.float32) -> torch.Tensor:
        d = self._bufs.setdefault(n, {})
        t = d.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            d[key] = t
        return t

    def _norm_consts(self):
        if self.norm is None:
            return None, None
        return [float(v) for v in self.norm.mean], [float(v) for v in self.norm.denom]

    # --------------------------------------------------------------- input
    def _host_augmentation(self, n: int) -> np.ndarray:
        flip = (self.np_rng.uniform(size=n) <= 0.5).astype(np.float32)
        ang = self.np_rng.uniform(-0.05, 0.05, size=n) * 2.0 * math.pi
        ct = self.np_rng.uniform(0.9, 1.1, size=n).astype(np.float32)
        return np.stack([flip, np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32), ct], 1)

    def draw_augmentation(self, n: int) -> torch.Tensor:
        """Per-image {flip, cos, sin, contrast}: RandomFlip("horizontal"), RandomRotation(0.05)
        (angle ~ U(-0.05, 0.05) * 2 pi), RandomContrast(0.1) (factor ~ U(0.9, 1.1)); cnn.py:76-80."""
        a = self._host_augmentation(n)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def _host_dropout(self, n: int, out: List[torch.Tensor]) -> None:
        """Keep-scales written into the host views `out` (one per stage, then the head)."""
        k = 0
        if self.drop_block > 0:
            for f in self.widths:
                keep = (torch.rand((n, f), generator=self.gen) >= self.drop_block).float()
                torch.div(keep, 1.0 - self.drop_block, out=out[k])
                k += 1
        if self.drop_top > 0:
            keep = (torch.rand((n, self.widths[-1]), generator=self.gen) >= self.drop_top).float()
            torch.div(keep, 1.0 - self.drop_top, out=out[k])

    def draw_step_randoms(self, n: int, augment: bool):
        """All per-step host draws (SpatialDropout2D / Dropout keep-scales, in-model
        augmentation parameters) through ONE pinned staging buffer and one asynchronous copy:
        pageable uploads would block the host until the stream drains, once per tensor.
        Returns (drops, top_drop, aug4) as views of a persistent device buffer."""
        sizes = []
        if self.drop_block > 0:
            sizes += [(n, f) for f in self.widths]
        if self.drop_top > 0:
            sizes.append((n, self.widths[-1]))
        if augment:
            sizes.append((n, 4))
        if not sizes:
            return None, None, None
        st = self._stage.get((n, augment))
        if st is None:
            total = sum(a * b for a, b in sizes)
            st = {"dev": torch.empty(total, dtype=torch.float32, device=self.device),
                  "ring": [(torch.empty(total, dtype=torch.float32).pin_memory(),
                            torch.cuda.Event()) for _ in range(3)], "i": 0, "used": 0}
            self._stage[(n, augment)] = st

        def views(flat):
            out, off = [], 0
            for a, b in sizes:
                out.append(flat[off:off + a * b].view(a, b))
                off += a * b
            return out

        host, ev = st["ring"][st["i"]]
        if st["used"] >= len(st["ring"]):
            ev.synchronize()  # the copy that last read this pinned slot has executed
        st["i"] = (st["i"] + 1) % len(st["ring"])
        st["used"] += 1
        hv = views(host)
        self._host_dropout(n, hv)
        if augment:
            hv[-1].copy_(torch.from_numpy(self._host_augmentation(n)))
        st["dev"].copy_(host, non_blocking=True)
        ev.record()
        dv = views(st["dev"])
        k = 0
        drops = top = aug4 = None
        if self.drop_block > 0:
            drops = dv[:len(self.widths)]
            k = len(self.widths)
        if self.drop_top > 0:
            top = dv[k]
        if augment:
            aug4 = dv[-1]
        return drops, top, aug4

    def _input(self, x, training: bool, aug4: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Accepts uint8 [N,H,W,3] (host or device) or float32 [N,H,W,3] in [0,1] (the
        reference loader's format); returns normalised f32 NCHW on the device."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        x = x.to(self.device, non_blocking=True)
        n = x.shape[0]
        mean, denom = self._norm_consts()
        out = self._buf(n, "x0", (n, 3, x.shape[1], x.shape[2]))
        if x.dtype == torch.uint8:
            x = x.contiguous()
            if training and self.augment:
                if aug4 is None:
                    aug4 = self.draw_augmentation(n)
                return nn.input_stage(x, aug4, mean, denom, out=out)
            return ops.pack_hwc_u8_to_nchw_f32(x, mean, denom, out=out)
        if x.dtype != torch.float32:
            raise TypeError("model input must be uint8 or float32 [N,H,W,3]")
        if training and self.augment:
            u8 = (x * 255.0).round().clamp_(0, 255).to(torch.uint8).contiguous()
            if aug4 is None:
                aug4 = self.draw_augmentation(n)
            return nn.input_stage(u8, aug4, mean, denom, out=out)
        xc = x.permute(0, 3, 1, 2).contiguous()
        if mean is None:
            return xc
        sc = torch.tensor([1.0 / d for d in denom], dtype=torch.float32, device=self.device)
        sh = torch.tensor([-m / d for m, d in zip(mean, denom)], dtype=torch.float32,
                          device=self.device)
        return nn.scale_shift_act(xc, sc, sh, False, out=out)

    # ------------------------------------------------------------- forward
    def _bn(self, name: str, y: torch.Tensor, training: bool) -> torch.Tensor:
        st = self.stats[name]
        if training:
            nn.bn_train_stats(y, self.p[name + ".gamma"], self.p[name + ".beta"],
                              self.s[name + ".mean"], self.s[name + ".var"], st, BN_MOMENTUM, BN_EPS)
        else:
            nn.bn_infer_scale_shift(self.p[name + ".gamma"], self.p[name + ".beta"],
                                    self.s[name + ".mean"], self.s[name + ".var"], st, BN_EPS)
        return st

    def _conv_bn(self, x, wname: str, ksize: int, bn: str, pro, out: torch.Tensor, training: bool):
        """Conv2D -> BatchNormalization: returns (y, stats[4,C]).  In training the batch
        statistics come out of the convolution's epilogue (no second pass over y)."""
        P = self.p
        if training:
            st = self.stats[bn]
            nn.conv2d_bn_stats(x, P[wname], ksize, P[bn + ".gamma"], P[bn + ".beta"],
                               self.s[bn + ".mean"], self.s[bn + ".var"], st, pro[0], pro[1], pro[2],
                               out=out, momentum=BN_MOMENTUM, eps=BN_EPS)
            return out, st
        w = P[wname]
        if self.infer_dtype == "bf16" and x.shape[3] % 4 == 0 and w.shape[2] % 32 == 0:
            # reduced-precision inference: bf16 operands, fp32 accumulation (packed weights are
            # rebuilt per call: 1.25 M parameters, microseconds)
            y = nn.conv2d_bf16(x, nn.conv2d_bf16_weights(w, ksize), w.shape[2], ksize, pro[0], pro[1],
                               pro[2], out=out)
        else:
            y = nn.conv2d(x, w, ksize, pro[0], pro[1], pro[2], out=out)
        return y, self._bn(bn, y, False)

    def _bf16_storage_ok(self, h: int, w: int) -> bool:
        """The bf16-activation forward needs 4-pixel groups at every stage (conv staging, the
        2x2-pool tail, the final mean) and 32-channel output blocks."""
        for f in self.widths:
            if f % 32 or w % 4 or h % 2:
                return False
            h, w = h // 2, w // 2
        return (h * w) % 4 == 0

    def _forward_infer_bf16(self, x0: torch.Tensor) -> torch.Tensor:
        """Inference forward with bf16 conv operands AND bf16 activation storage (the layers'
        outputs, as Keras' mixed_float16 keeps them).  Each convolution applies its folded
        BatchNorm (+ReLU) to the fp32 accumulators before rounding, so what is stored IS the
        activation and the next convolution stages the stored bits without arithmetic; SE gate,
        residual add, pooling means and the dense head compute in fp32.  Returns probs [N, C]."""
        n, _c, _h, _w = x0.shape
        P, bf = self.p, torch.bfloat16

        # Packed bf16 weights and folded BatchNorm coefficients are kept for as long as the parameters stand:
        # torch's version counters see every in-place write through torch (set_weights, restored best weights, EMA
        # swaps, a test poking a tensor), `_mut` counts the passes whose kernels write them (training forward,
        # optimizer step).  96 small launches per forward pass otherwise (~0.5 ms at batch 1,024, most of the
        # time of a one-image predict).
        key = (self.flat_p._version, self.flat_s._version, self._mut)
        if self._infer_cache.get("key") != key:
            self._infer_cache = {"key": key, "w": {}, "bn": {}}
        cache = self._infer_cache

        def conv(x, wname, k, bn, relu, means=None):
            wt = P[wname]
            wp = cache["w"].get(wname)
            if wp is None:
                wp = cache["w"][wname] = nn.conv2d_bf16_weights(wt, k)
            st = cache["bn"].get(bn)
            if st is None:   # inference: scale / shift from the moving statistics (own copy: `stats` is shared)
                st = cache["bn"][bn] = self._bn(bn, None, False).clone()
            out = self._buf(n, "bf16." + wname, (n, wt.shape[2], x.shape[2], x.shape[3]), bf)
            if means is not None:   # + the squeeze of the block's SE gate, summed in the epilogue
                return nn.conv2d_bf16_mean(x, wp, wt.shape[2], k, out, means, out_scale=st[2], out_shift=st[3],
                                           out_relu=relu)[0]
            return nn.conv2d_bf16(x, wp, wt.shape[2], k, out=out, out_scale=st[2], out_shift=st[3], out_relu=relu)

        xin = conv(x0, "stem.w", 3, "stem.bn", True)
        cin = self.widths[0]
        for i, f in enumerate(self.widths):
            p = f"s{i}."
            a1 = conv(xin, p + "c1.w", 3, p + "bn1", True)
            m = self._buf(n, p + "m", (n, f)) if self.use_se else None
            a2 = conv(a1, p + "c2.w", 3, p + "bn2", True, means=m)
            s = None
            if self.use_se:
                s = nn.se_fwd(m, P[p + "se.w1"], P[p + "se.b1"], P[p + "se.w2"], P[p + "se.b2"],
                              self._buf(n, p + "z1", (n, f // 8)), self._buf(n, p + "s", (n, f)))
            sc = conv(xin, p + "proj.w", 1, p + "bnp", False) if cin != f else xin
            xin = nn.block_tail_fwd_bf16(a2, None, None, s, sc, None, None, False,
                                         out=self._buf(n, "bf16." + p + "p", (n, f, a2.shape[2] // 2,
                                                                              a2.shape[3] // 2), bf))
            cin = f
        g = nn.gap_bf16(xin, out=self._buf(n, "g", (n, self.widths[-1])))
        probs = self._buf(n, "probs", (n, self.num_classes))
        nn.head_fwd(g, P["dense.w"], P["dense.b"], None, probs, None)
        return probs

    def set_inference_dtype(self, dtype: str) -> None:
        """"f32" (default) or "bf16": the arithmetic of the convolutions in predict / evaluate.
        The reference runs them in half precision under its default mixed_float16 policy
        (train.py:53-117).  The training step's precision is set_training_dtype's."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(f"inference dtype must be 'f32' or 'bf16', got {dtype!r}")
        self.infer_dtype = dtype

    def set_training_dtype(self, dtype: str) -> None:
        """"f32" (default) or "bf16": the mixed-precision training step (the reference's default
        policy is mixed_float16, train.py:179-190; `--no-mixed-precision` selects f32).  bf16:
        activations and gradients are STORED as bf16 and every convolution operand is bf16
        (fp32 accumulation); master weights, Adam state, BatchNorm statistics, SE, softmax and the
        loss stay fp32.  Needs widths that are multiples of 32 and an image size that keeps every
        stage a multiple of four pixels wide (224, 64, 32 do)."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(f"training dtype must be 'f32' or 'bf16', got {dtype!r}")
        if dtype == "bf16" and not (self.use_se and self._bf16_storage_ok(self.img_size, self.img_size)):
            raise ValueError("bf16 training needs use_se, widths % 32 == 0 and every stage a multiple "
                             f"of 4 pixels wide (img_size {self.img_size}, widths {self.widths})")
        self.train_dtype = dtype

    # ------------------------------------------------- mixed-precision step (bf16 storage)
    def _prep_bf16_weights(self) -> None:
        """bf16 copies of the convolution kernels in MFMA operand order, for the forward and the
        input-gradient convolutions: rebuilt from the fp32 masters every step (1.25 M values)."""
        for name, shape, kind in self.specs:
            if len(shape) != 3:
                continue
            k = 3 if shape[1] == 9 else 1
            self._wt["f:" + name] = nn.conv2d_bf16_weights(self.p[name], k)
            if name != "stem.w":
                self._wt["d:" + name] = nn.conv2d_bf16_dgrad_weights(self.p[name], k)

    def _forward_train_bf16(self, x0: torch.Tensor, y_true: torch.Tensor, drops, top_drop):
        """The training forward pass on bf16 storage; everything backward needs goes to self._saved."""
        n, _c, h, w = x0.shape
        self._mut += 1
        P, bf = self.p, torch.bfloat16
        B = lambda k, shape, dt=bf: self._buf(n, "t16." + k, shape, dt)  # noqa: E731
        F32 = torch.float32
        sv: Dict[str, Any] = {"x0": x0, "n": n}
        self._prep_bf16_weights()

        def conv_bn(x, wname, k, bn, pro, out):
            cout = P[wname].shape[2]
            st = self.stats[bn]
            nn.conv2d_bn_stats_bf16(x, self._wt["f:" + wname], cout, k, P[bn + ".gamma"], P[bn + ".beta"],
                                    self.s[bn + ".mean"], self.s[bn + ".var"], st, pro[0], pro[1], pro[2],
                                    out=out, momentum=BN_MOMENTUM, eps=BN_EPS)
            return out, st

        y, st = conv_bn(x0, "stem.w", 3, "stem.bn", (None, None, False), B("stem.y", (n, self.widths[0], h, w)))
        sv["stem.y"] = y
        xin, xin_st, cin = y, st, self.widths[0]
        for i, f in enumerate(self.widths):
            p = f"s{i}."
            pro = (xin_st[2], xin_st[3], True) if xin_st is not None else (None, None, False)
            y1, st1 = conv_bn(xin, p + "c1.w", 3, p + "bn1", pro, B(p + "y1", (n, f, h, w)))
            y2, st2 = conv_bn(y1, p + "c2.w", 3, p + "bn2", (st1[2], st1[3], True), B(p + "y2", (n, f, h, w)))
            msum = B(p + "msum", (n, f, 2), F32)
            m = nn.gap_stats_bf16(y2, out=B(p + "m", (n, f), F32), scale=st2[2], shift=st2[3], relu=True,
                                  mask_sums=msum)
            z1 = B(p + "z1", (n, f // 8), F32)
            s = nn.se_fwd(m, P[p + "se.w1"], P[p + "se.b1"], P[p + "se.w2"], P[p + "se.b2"], z1,
                          B(p + "s", (n, f), F32))
            sv[p + "msum"], sv[p + "m"], sv[p + "z1"] = msum, m, z1
            if cin != f:
                yp, stp = conv_bn(xin, p + "proj.w", 1, p + "bnp", pro, B(p + "yp", (n, f, h, w)))
                sc, scs, scb, scr = yp, stp[2], stp[3], False
                sv[p + "yp"] = yp
            else:
                sc, scs, scb, scr = xin, pro[0], pro[1], pro[2]
            drop = drops[i] if drops is not None else None
            pooled = B(p + "p", (n, f, h // 2, w // 2))
            route = B(p + "route", pooled.shape, torch.uint8)
            nn.block_tail_fwd_train_bf16(y2, st2[2], st2[3], s, sc, scs, scb, scr, drop, route, pooled)
            sv.update({p + "xin": xin, p + "xin_st": xin_st, p + "y1": y1, p + "y2": y2, p + "s": s,
                       p + "route": route, p + "drop": drop, p + "hw": (h, w)})
            xin, xin_st, cin, h, w = pooled, None, f, h // 2, w // 2
        g = nn.gap_stats_bf16(xin, out=B("g", (n, self.widths[-1]), F32))
        feat = g
        if top_drop is not None:
            feat = nn.mul(g, top_drop, B("feat", g.shape, F32))
        probs = B("probs", (n, self.num_classes), F32)
        loss = B("loss", (n,), F32)
        nn.head_fwd(feat, P["dense.w"], P["dense.b"], y_true, probs, loss)
        sv.update({"feat": feat, "top_drop": top_drop, "probs": probs, "y_true": y_true, "last_hw": (h, w)})
        self._saved = sv
        return probs, loss

    def _backward_bf16(self, part: Optional[int] = None) -> None:
        """Fills flat_g (fp32) from the tensors of the last _forward_train_bf16.  part: see backward()."""
        sv, P, G = self._saved, self.p, self.g
        n = sv["n"]
        bf, F32 = torch.bfloat16, torch.float32
        B = lambda k, shape, dt=bf: self._buf(n, "t16." + k, shape, dt)  # noqa: E731
        f_last = self.widths[-1]
        if part in (None, 0):
            dlogits = B("dlogits", (n, self.num_classes), F32)
            dfeat = B("dfeat", (n, f_last), F32)
            nn.head_bwd(sv["feat"], P["dense.w"], sv["probs"], sv["y_true"], dlogits, dfeat,
                        G["dense.w"], G["dense.b"], 1.0 / (self._global_n or n))
            dg = dfeat
            if sv["top_drop"] is not None:
                dg = nn.mul(dfeat, sv["top_drop"], B("dg", dfeat.shape, F32))
            h, w = sv["last_hw"]
            dp = nn.bcast_planes_bf16(dg, h, w, 1.0 / (h * w), B("dp_last", (n, f_last, h, w)))
        else:
            dp = sv["bwd_dp"]
        for i in self._backward_stages(part):
            f = self.widths[i]
            cin = self.widths[i - 1] if i > 0 else self.widths[0]
            p = f"s{i}."
            h, w = sv[p + "hw"]
            xin, xin_st, y1, y2 = (sv[p + k] for k in ("xin", "xin_st", "y1", "y2"))
            pro = (xin_st[2], xin_st[3], True) if xin_st is not None else (None, None, False)
            s, route, drop = sv[p + "s"], sv[p + "route"], sv[p + "drop"]
            st1, st2 = self.stats[p + "bn1"], self.stats[p + "bn2"]
            gA, gB, gC = B(p + "gA", y1.shape), B(p + "gB", y1.shape), B(p + "gC", y1.shape)
            ds = B(p + "ds", (n, f), F32)
            psum = B(p + "psum", (n, f, 2), F32)
            yp = sv.get(p + "yp")
            psum_p = B(p + "psum_p", (n, f, 2), F32) if yp is not None else None
            # dr (-> gA), the SE gate gradient and BN2's per-plane backward sums in one pass
            nn.block_tail_bwd_bf16(dp, route, y2, st2[2], st2[3], drop, gA, ds, psum, yp, psum_p)
            dm = B(p + "dm", (n, f), F32)
            nn.se_bwd(ds, sv[p + "m"], sv[p + "z1"], s, P[p + "se.w1"], P[p + "se.w2"], dm,
                      G[p + "se.w1"], G[p + "se.b1"], G[p + "se.w2"], G[p + "se.b2"], dm_scale=1.0 / (h * w))
            # BN2 backward + conv2 weight gradient; dy2 -> gB
            nn.bn_bwd_wgrad_bf16(y1, gA, y2, st2, P[p + "bn2.gamma"], G[p + "bn2.gamma"], G[p + "bn2.beta"],
                                 True, 3, G[p + "c2.w"], gB, st1[2], st1[3], True, alpha_nc=s, add_nc=dm,
                                 plane_g=psum, plane_m=sv[p + "msum"])
            # da1 -> gC; the epilogue leaves BN1's backward sums
            _, tsum = nn.conv2d_bf16_train(gB, self._wt["d:" + p + "c2.w"], f, 3, gC, mask_y=y1,
                                           mask_scale=st1[2], mask_shift=st1[3], mask_relu=True)
            # BN1 backward + conv1 weight gradient; dy1 -> gB
            nn.bn_bwd_wgrad_bf16(xin, gC, y1, st1, P[p + "bn1.gamma"], G[p + "bn1.gamma"], G[p + "bn1.beta"],
                                 True, 3, G[p + "c1.w"], gB, pro[0], pro[1], pro[2], tile_sums=tsum)
            if cin != f:
                stp = self.stats[p + "bnp"]
                # projection BN backward + 1x1 weight gradient; dyp -> gC
                nn.bn_bwd_wgrad_bf16(xin, gA, yp, stp, P[p + "bnp.gamma"], G[p + "bnp.gamma"],
                                     G[p + "bnp.beta"], False, 1, G[p + "proj.w"], gC, pro[0], pro[1], pro[2],
                                     plane_g=psum_p)
                dx = B(p + "dx", xin.shape)
                nn.conv2d_bf16_train(gC, self._wt["d:" + p + "proj.w"], cin, 1, dx)
            else:
                dx = gA  # identity shortcut: dx starts as dr
            stem_sums = None
            if i == 0:  # dx feeds the stem's BN backward: gather its sums in this epilogue
                _, stem_sums = nn.conv2d_bf16_train(gB, self._wt["d:" + p + "c1.w"], cin, 3, dx, accumulate=True,
                                                    mask_y=sv["stem.y"], mask_scale=self.stats["stem.bn"][2],
                                                    mask_shift=self.stats["stem.bn"][3], mask_relu=True)
            else:
                nn.conv2d_bf16_train(gB, self._wt["d:" + p + "c1.w"], cin, 3, dx, accumulate=True)
            dp = dx
        if part == 0:
            sv["bwd_dp"] = dp   # the gradient that enters stage 0: where part 1 picks up
            return
        # the stem has no input gradient: its BN backward exists only inside the wgrad kernel
        nn.bn_bwd_wgrad_bf16(sv["x0"], dp, sv["stem.y"], self.stats["stem.bn"], P["stem.bn.gamma"],
                             G["stem.bn.gamma"], G["stem.bn.beta"], True, 3, G["stem.w"], None,
                             tile_sums=stem_sums)

    def forward(self, x0: torch.Tensor, training: bool, y_true: Optional[torch.Tensor] = None,
                drops: Optional[List[torch.Tensor]] = None,
                top_drop: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """x0: normalised f32 NCHW.  Returns (probs [N,C], per-sample loss or None).
        In training mode every tensor the backward pass needs is kept in self._saved."""
        n, _c, h, w = x0.shape
        if training:
            self._mut += 1   # the BatchNorm layers update their moving statistics
        P, B = self.p, lambda k, shape: self._buf(n, k, shape)
        sv: Dict[str, Any] = {"x0": x0, "n": n}
        # Activations a = relu(BN(y)) are never materialised: every consumer (the next conv,
        # wgrad, GAP, the residual tail, BN backward) applies scale/shift(+ReLU) while it reads y.
        y, st = self._conv_bn(x0, "stem.w", 3, "stem.bn", (None, None, False),
                              B("stem.y", (n, self.widths[0], h, w)), training)
        sv["stem.y"] = y
        xin, xin_st = y, st  # block input = relu(xin*xin_st[2]+xin_st[3]) (None = already final)
        cin = self.widths[0]
        for i, f in enumerate(self.widths):
            p = f"s{i}."
            pro = (xin_st[2], xin_st[3], True) if xin_st is not None else (None, None, False)
            y1, st1 = self._conv_bn(xin, p + "c1.w", 3, p + "bn1", pro, B(p + "y1", (n, f, h, w)),
                                    training)
            y2, st2 = self._conv_bn(y1, p + "c2.w", 3, p + "bn2", (st1[2], st1[3], True),
                                    B(p + "y2", y1.shape), training)
            s = None
            if self.use_se:
                # the squeeze pass also leaves BN2's ReLU-mask sums for the backward pass
                msum = B(p + "msum", (n, f, 2)) if training else None
                m = nn.gap(y2, out=B(p + "m", (n, f)), scale=st2[2], shift=st2[3], relu=True,
                           mask_sums=msum)
                sv[p + "msum"] = msum
                z1 = B(p + "z1", (n, f // 8))
                s = nn.se_fwd(m, P[p + "se.w1"], P[p + "se.b1"], P[p + "se.w2"], P[p + "se.b2"], z1,
                              B(p + "s", (n, f)))
                sv[p + "m"], sv[p + "z1"] = m, z1
            if cin != f:
                yp, stp = self._conv_bn(xin, p + "proj.w", 1, p + "bnp", pro, B(p + "yp", y1.shape),
                                        training)
                sc, scs, scb, scr = yp, stp[2], stp[3], False
                sv[p + "yp"] = yp
            else:
                sc, scs, scb, scr = xin, pro[0], pro[1], pro[2]
            drop = drops[i] if (training and drops is not None) else None
            pooled = B(p + "p", (n, f, h // 2, w // 2))
            route = self._buf(n, p + "route", pooled.shape, torch.uint8)
            nn.block_tail_fwd(y2, st2[2], st2[3], s, sc, scs, scb, scr, drop, route, pooled)
            sv.update({p + "xin": xin, p + "xin_st": xin_st, p + "y1": y1, p + "y2": y2, p + "s": s,
                       p + "route": route, p + "drop": drop, p + "hw": (h, w)})
            xin, xin_st, cin, h, w = pooled, None, f, h // 2, w // 2
        a = xin
        g = nn.gap(a, out=B("g", (n, self.widths[-1])))
        feat = g
        if training and top_drop is not None:
            feat = nn.mul(g, top_drop, B("feat", g.shape))
        probs = B("probs", (n, self.num_classes))
        loss = B("loss", (n,)) if y_true is not None else None
        nn.head_fwd(feat, P["dense.w"], P["dense.b"], y_true, probs, loss)
        sv.update({"feat": feat, "top_drop": top_drop, "probs": probs, "y_true": y_true,
                   "last_hw": (h, w)})
        if training:
            self._saved = sv
        return probs, loss

    # ------------------------------------------------------------ backward
    def _backward_stages(self, part: Optional[int]):
        """Stage indices a backward pass walks, last stage first.  part None: all of them.  The data-parallel step
        runs the pass in two parts so that the gradients of the first part cross xGMI while the second computes:
        part 0 = head and stages >= 1 (98 % of the parameters: everything from `grad_split()` on in the flat
        bucket), part 1 = stage 0 and the stem (the 224 x 224 layers: a large share of the time, 2 % of the
        parameters)."""
        last = len(self.widths) - 1
        if part is None:
            return range(last, -1, -1)
        return range(last, 0, -1) if part == 0 else range(0, -1, -1)

    def grad_split(self) -> int:
        """Offset in the flat parameter / gradient buffers of the first tensor that does NOT belong to the stem or
        stage 0: [0, split) is written by backward part 1, [split, n_params) by part 0 (train_step rounds it UP to
        whole cache lines for the exchange: the few part-0 elements below the rounded offset go with the second
        piece, by which time they are long complete)."""
        for (name, _s, _k), off in zip(self.specs, self.offsets.tolist()):
            if not (name.startswith("stem.") or name.startswith("s0.")):
                return int(off)
        return self.n_params

    def backward(self, part: Optional[int] = None) -> None:
        """Fills flat_g with d(mean data loss)/d(param) for the last training forward (part: _backward_stages)."""
        sv, P, G = self._saved, self.p, self.g
        n = sv["n"]
        B = lambda k, shape: self._buf(n, k, shape)  # noqa: E731
        f_last = self.widths[-1]
        if part in (None, 0):
            dlogits = B("dlogits", (n, self.num_classes))
            dfeat = B("dfeat", (n, f_last))
            nn.head_bwd(sv["feat"], P["dense.w"], sv["probs"], sv["y_true"], dlogits, dfeat,
                        G["dense.w"], G["dense.b"], 1.0 / (self._global_n or n))
            dg = dfeat
            if sv["top_drop"] is not None:
                dg = nn.mul(dfeat, sv["top_drop"], B("dg", dfeat.shape))
            h, w = sv["last_hw"]
            dp = nn.bcast_planes(dg, h, w, 1.0 / (h * w), out=B("dp_last", (n, f_last, h, w)))
        else:
            dp = sv["bwd_dp"]
        for i in self._backward_stages(part):
            f = self.widths[i]
            cin = self.widths[i - 1] if i > 0 else self.widths[0]
            p = f"s{i}."
            h, w = sv[p + "hw"]
            xin, xin_st, y1, y2 = (sv[p + k] for k in ("xin", "xin_st", "y1", "y2"))
            pro = (xin_st[2], xin_st[3], True) if xin_st is not None else (None, None, False)
            s, route, drop = sv[p + "s"], sv[p + "route"], sv[p + "drop"]
            st1, st2 = self.stats[p + "bn1"], self.stats[p + "bn2"]
            gA = B(p + "gA", y1.shape)
            gB = B(p + "gB", y1.shape)
            gC = B(p + "gC", y1.shape)
            ds = B(p + "ds", (n, f)) if self.use_se else None
            # dr (into gA), the SE gate gradient, and BN2's per-plane backward sums in one pass
            psum = B(p + "psum", (n, f, 2))
            yp = sv.get(p + "yp")  # projection shortcut: its BN's backward sums ride along
            psum_p = B(p + "psum_p", (n, f, 2)) if yp is not None else None
            nn.block_tail_bwd(dp, route, y2, st2[2], st2[3], drop, gA, ds, psum, yp, psum_p)
            add_nc = None
            if self.use_se:
                dm = B(p + "dm", (n, f))
                nn.se_bwd(ds, sv[p + "m"], sv[p + "z1"], s, P[p + "se.w1"], P[p + "se.w2"], dm,
                          G[p + "se.w1"], G[p + "se.b1"], G[p + "se.w2"], G[p + "se.b2"],
                          dm_scale=1.0 / (h * w))
                add_nc = dm
            # conv2 branch: dz2 = (dr*s + dm/HW) * [a2 > 0];
            # BN2 backward + conv2 weight gradient: dy2 is formed inside the wgrad kernel (-> gB)
            nn.bn_bwd_wgrad(y1, gA, y2, st2, P[p + "bn2.gamma"], G[p + "bn2.gamma"],
                            G[p + "bn2.beta"], True, 3, G[p + "c2.w"], gB, st1[2], st1[3], True,
                            alpha_nc=s, add_nc=add_nc, plane_g=psum,
                            plane_m=sv[p + "msum"] if self.use_se else None)
            # da1 (-> gC); its epilogue leaves BN1's backward sums
            _, tsum = nn.conv2d_bnbwd(gB, self._dgrad_w(p + "c2.w", 3), 3, y1, st1, True, gC)
            # BN1 backward + conv1 weight gradient (dy1 -> gB)
            nn.bn_bwd_wgrad(xin, gC, y1, st1, P[p + "bn1.gamma"], G[p + "bn1.gamma"],
                            G[p + "bn1.beta"], True, 3, G[p + "c1.w"], gB, pro[0], pro[1], pro[2],
                            tile_sums=tsum)
            if cin != f:
                stp = self.stats[p + "bnp"]
                # projection BN backward + 1x1 weight gradient (dyp -> gC)
                nn.bn_bwd_wgrad(xin, gA, yp, stp, P[p + "bnp.gamma"], G[p + "bnp.gamma"],
                                G[p + "bnp.beta"], False, 1, G[p + "proj.w"], gC, pro[0], pro[1], pro[2],
                                plane_g=psum_p)
                dx = B(p + "dx", xin.shape)
                nn.conv2d(gC, self._dgrad_w(p + "proj.w", 1), 1, out=dx)
            else:
                dx = gA  # identity shortcut: dx starts as dr
            stem_sums = None
            if i == 0:  # dx feeds the stem's BN backward: gather its sums in this epilogue
                _, stem_sums = nn.conv2d_bnbwd(gB, self._dgrad_w(p + "c1.w", 3), 3, sv["stem.y"],
                                               self.stats["stem.bn"], True, dx, accumulate=True)
            else:
                nn.conv2d(gB, self._dgrad_w(p + "c1.w", 3), 3, out=dx, accumulate=True)
            dp = dx
        if part == 0:
            sv["bwd_dp"] = dp   # the gradient that enters stage 0: where part 1 picks up
            return
        # stem: dp is the gradient wrt relu(BN(stem.y))
        # the stem has no input gradient: its BN backward exists only inside the wgrad kernel
        nn.bn_bwd_wgrad(sv["x0"], dp, sv["stem.y"], self.stats["stem.bn"], P["stem.bn.gamma"],
                        G["stem.bn.gamma"], G["stem.bn.beta"], True, 3, G["stem.w"], None,
                        tile_sums=stem_sums)

    def _dgrad_w(self, name: str, k: int) -> torch.Tensor:
        return nn.conv2d_dgrad_weights(self.p[name], k)

    # ------------------------------------------------------------ training
    def draw_dropout(self, n: int):
        """SpatialDropout2D keep-scales [N,C_i] per stage and Dropout keep-scales [N,F]."""
        host = []
        if self.drop_block > 0:
            host += [torch.empty((n, f)) for f in self.widths]
        if self.drop_top > 0:
            host.append(torch.empty((n, self.widths[-1])))
        self._host_dropout(n, host)
        drops = [t.to(self.device) for t in host[:len(self.widths)]] if self.drop_block > 0 else None
        top = host[-1].to(self.device) if self.drop_top > 0 else None
        return drops, top

    def train_step(self, x, y_true: Optional[torch.Tensor], lr: float, *, weight_decay: float = 1e-4,
                   clipnorm: float = 0.5, ema_decay: float = 0.999, adamw: bool = True,
                   grad_sync=None, global_n: Optional[int] = None, grad_overlap=None):
        """One optimisation step on a batch.  y_true: f32 [N,C] (already label-smoothed).
        grad_sync(flat_g) is called between backward and the optimizer (data-parallel
        all-reduce); with global_n the local gradient is scaled by 1/global_n so that a SUM
        all-reduce yields the global-batch mean.  Returns (probs, per-sample loss) device
        tensors (no host sync).

        An EMPTY local batch (a data-parallel rank whose rank-strided slice of a ragged last
        global batch holds nothing) still takes the step: it contributes a zero gradient to the
        all-reduce and advances `opt_step`, Adam moments, weights and EMA exactly like its peers,
        so every rank issues the same collectives and the replicas stay bit-equal.  Returns
        (None, None) in that case.

        grad_overlap (a train.parallel.DataParallel with `overlap` on) replaces grad_sync: the backward pass is run
        in two parts and the all-reduce of the first part's gradients — head and stages >= 1, 98 % of the bucket —
        is in flight over xGMI while the second part (stage 0 and the stem, the 224 x 224 layers) computes; the
        small remainder follows.  Same kernels in the same order on the compute stream, same reductions: the
        parameters come out bit-equal to the non-overlapped step."""
        self._mut += 1   # parameters and moving statistics change (also when the step is a graph replay)
        n = int(x.shape[0])
        self._global_n = global_n
        if grad_overlap is not None:
            # two exchanges, the same two on every rank (a rank with an empty batch sends zeros in both)
            split = min(self.n_params, -(-self.grad_split() // 64) * 64)   # whole 256-byte lines per piece
            handles = []
            if n == 0:
                self.flat_g.zero_()
                probs = loss = None
                handles.append(grad_overlap.allreduce_begin(self.flat_g, split, self.n_params))
            else:
                probs, loss = self._forward_backward(x, y_true, between=lambda: handles.append(
                    grad_overlap.allreduce_begin(self.flat_g, split, self.n_params)))
            handles.append(grad_overlap.allreduce_begin(self.flat_g, 0, split))
            grad_overlap.allreduce_finish(self.flat_g, handles)
            grad_sync = None
        elif n == 0:
            self.flat_g.zero_()
            probs = loss = None
        else:
            probs, loss = self._forward_backward(x, y_true)
        if grad_sync is not None:
            grad_sync(self.flat_g)
        self.opt_step += 1
        self._optimizer_update(lr, weight_decay=weight_decay if adamw else 0.0, clipnorm=clipnorm,
                               ema_decay=ema_decay)
        return probs, loss

    def _forward_backward(self, x, y_true: torch.Tensor, between=None):
        """Forward + backward of one local batch: fills flat_g, returns (probs, loss).  `between`, when given, is
        called on the host between the two parts of the backward pass (backward()'s `part`): everything of part 0
        has been issued to the stream by then, nothing of part 1.

        The ~250 launches of a step are recorded once per (batch shape, precision) into a HIP graph
        and replayed (the host would otherwise spend 3-6 ms per step issuing them, a fifth of the
        bf16 step): the first two steps of a shape run eagerly (they size every buffer), the third is
        captured.  Inputs and labels are copied into fixed buffers, the per-step random draws go up
        through their fixed staging buffer before the replay; the optimizer (its learning rate
        changes every step) and the gradient all-reduce stay outside.  LEAFFLICTION_GRAPH=0 turns
        this off."""
        if self._graphs_on and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.uint8:
            return self._forward_backward_graph(x, y_true, between)
        return self._forward_backward_eager(x, y_true, between)

    def _forward_backward_graph(self, x: torch.Tensor, y_true: torch.Tensor, between=None):
        n = int(x.shape[0])
        key = (tuple(x.shape), self.train_dtype, self.augment, self._global_n, tuple(y_true.shape),
               between is not None)
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"calls": 0, "graph": None}
        st["calls"] += 1
        if st["graph"] is not None and st["ws_gen"] != nn.workspace_generation():
            # a launch somewhere in the process (a larger batch, another model) has replaced one of the library's
            # workspace buffers since this graph was recorded: its nodes hold pointers into the freed buffer.
            # Drop it and record the step again (this call runs eagerly on the current buffers).
            st["graph"], st["calls"] = None, 2
        if st["graph"] is None and st["calls"] <= 2:
            return self._forward_backward_eager(x, y_true, between)
        drops, top, aug4 = self.draw_step_randoms(n, self.augment)  # fixed device views, fresh values
        if st["graph"] is None:
            st["x"] = torch.empty_like(x)
            st["y"] = torch.empty_like(y_true)
            st["x"].copy_(x)
            st["y"].copy_(y_true)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            g2 = torch.cuda.CUDAGraph() if between is not None else None
            try:
                # with `between`: two graphs out of one memory pool, cut where the host may start the first exchange
                with torch.cuda.graph(g):
                    st["out"] = self._forward_backward_body(st["x"], st["y"], drops, top, aug4,
                                                            part=None if between is None else 0)
                if g2 is not None:
                    with torch.cuda.graph(g2, pool=g.pool()):
                        self._forward_backward_body(st["x"], st["y"], drops, top, aug4, part=1)
            except Exception:
                self._graphs_on = False   # capture is an optimisation: fall back to eager launches
                torch.cuda.synchronize()
                return self._forward_backward_split(x, y_true, drops, top, aug4, between)
            st["graph"], st["graph2"] = g, g2
            st["ws_gen"] = nn.workspace_generation()
        else:
            st["x"].copy_(x)
            st["y"].copy_(y_true)
        st["graph"].replay()
        if between is not None:
            between()
            st["graph2"].replay()
        return st["out"]

    def _forward_backward_eager(self, x, y_true: torch.Tensor, between=None):
        n = int(x.shape[0])
        drops, top, aug4 = self.draw_step_randoms(n, self.augment)
        return self._forward_backward_split(x, y_true, drops, top, aug4, between)

    def _forward_backward_split(self, x, y_true, drops, top, aug4, between):
        if between is None:
            return self._forward_backward_body(x, y_true, drops, top, aug4)
        out = self._forward_backward_body(x, y_true, drops, top, aug4, part=0)
        between()
        self._forward_backward_body(x, y_true, drops, top, aug4, part=1)
        return out

    def _forward_backward_body(self, x, y_true, drops, top, aug4, part: Optional[int] = None):
        """part None: forward + the whole backward pass; 0: forward + backward part 0; 1: backward part 1."""
        bf16 = self.train_dtype == "bf16"
        if part == 1:
            self._backward_bf16(1) if bf16 else self.backward(1)
            return None
        x0 = self._input(x, True, aug4)
        if bf16:
            if not (self.use_se and self._bf16_storage_ok(x0.shape[2], x0.shape[3])):
                raise ValueError("bf16 training: unsupported shape (see set_training_dtype)")
            probs, loss = self._forward_train_bf16(x0, y_true, drops, top)
            self._backward_bf16(part)
        else:
            probs, loss = self.forward(x0, True, y_true, drops, top)
            self.backward(part)
        return probs, loss

    def _optimizer_update(self, lr: float, *, weight_decay: float, clipnorm: float,
                          ema_decay: float) -> None:
        nn.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v,
                      self.flat_ema if ema_decay > 0 else None, self.offsets, self.l2_vec,
                      self.max_count, lr, self.opt_step, weight_decay=weight_decay,
                      clipnorm=clipnorm, ema_decay=ema_decay, ema_copy=not self.ema_started,
                      norms=self.norms_ws)
        if ema_decay > 0:
            nn.ema_update(self.flat_s_ema, self.flat_s, ema_decay, not self.ema_started)
            self.ema_started = True

    def reseed_step_rng(self, seed: int) -> None:
        """Re-seed the per-step generators (SpatialDropout2D / Dropout masks, in-model
        augmentation draws).  Data-parallel ranks call this with seed + rank AFTER the weight
        broadcast: initial weights stay identical, the regularisation noise is independent
        across the shards of a global batch."""
        self.gen = torch.Generator(device="cpu").manual_seed(int(seed))
        self.np_rng = np.random.RandomState(int(seed) & 0x7FFFFFFF)

    def l2_penalty(self) -> torch.Tensor:
        tot = torch.zeros((), dtype=torch.float32, device=self.device)
        if self.l2_reg > 0:
            for name, _s, kind in self.specs:
                if kind == "w3":
                    tot = tot + self.l2_reg * (self.p[name] ** 2).sum()
        return tot

    # --------------------------------------------------------- keras-like API
    def compile(self, optimizer=None, loss=None, metrics=None) -> None:
        """optimizer: dict from train.utils.build_optimizer; loss: dict from build_loss."""
        self._compiled = {"optimizer": optimizer or {}, "loss": loss or {}, "metrics": metrics or []}

    def _targets(self, by) -> Tuple[torch.Tensor, torch.Tensor]:
        """labels (one-hot [B,C] or sparse [B]) -> (smoothed one-hot f32 on device, indices)."""
        if not isinstance(by, torch.Tensor):
            # pinned staging + asynchronous copy: a pageable upload would drain the stream
            by = torch.as_tensor(np.ascontiguousarray(by))
            if self.device.type == "cuda":
                by = by.pin_memory()
        by = by.to(self.device, non_blocking=True)
        if by.dim() == 1:
            idx = by.long()
            yt = torch.nn.functional.one_hot(idx, self.num_classes).float()
        else:
            yt = by.float()
            idx = yt.argmax(-1)
        ls = float(self._compiled.get("loss", {}).get("label_smoothing", 0.0) or 0.0)
        if ls > 0:
            yt = yt * (1.0 - ls) + ls / self.num_classes
        return yt.contiguous(), idx

    def fit(self, train_seq, validation_data=None, epochs: int = 1, callbacks=None, dp=None,
            verbose: int = 1):
        """Keras `Model.fit` for a ManifestSequence: per epoch the batch order is shuffled
        (keras shuffles Sequence batches), every batch is one `train_step`, then validation,
        callbacks and `train_seq.on_epoch_end()`.  `dp` is a train.parallel.DataParallel."""
        import logging
        import random as _random
        log = logging.getLogger(__name__)
        opt = self._compiled.get("optimizer", {})
        callbacks = callbacks or []
        hist: Dict[str, List[float]] = {}
        steps_per_epoch = len(train_seq)
        order_rng = _random.Random(12345)
        for cb in callbacks:
            cb.set_model(self)
            cb.on_train_begin()
        self.stop_training = False
        for epoch in range(epochs):
            order = list(range(steps_per_epoch))
            order_rng.shuffle(order)
            acc_loss = torch.zeros((), device=self.device)
            acc_correct = torch.zeros((), device=self.device)
            seen = 0
            prefetch = getattr(train_seq, "prefetch", None)
            for step_no, bi in enumerate(order):
                bx, by = train_seq[bi]
                if prefetch is not None and step_no + 1 < len(order):
                    prefetch(order[step_no + 1])   # its files decode on the codec workers during this step
                n_local = int(bx.shape[0])
                dp_on = dp is not None and dp.active
                if n_local == 0 and not dp_on:
                    continue
                # data-parallel: a rank with an empty slice of a ragged last global batch still
                # steps (zero gradient), or its peers would wait in the all-reduce forever
                yt, idx = self._targets(by) if n_local else (None, None)
                lr = opt["schedule"](self.opt_step) if "schedule" in opt else opt.get("lr", 1e-3)
                gn = train_seq.global_batch_size(bi) if dp_on else None
                probs, loss = self.train_step(
                    bx, yt, lr, weight_decay=opt.get("weight_decay", 0.0),
                    clipnorm=opt.get("clipnorm", 0.0), ema_decay=opt.get("ema_decay", 0.0),
                    adamw=opt.get("name", "adamw") == "adamw",
                    grad_sync=dp.allreduce_grads if dp_on else None, global_n=gn,
                    grad_overlap=dp if dp_on and getattr(dp, "overlap", False) else None)
                if n_local:
                    acc_loss += loss.sum()
                    acc_correct += (probs.argmax(-1) == idx).sum()
                    seen += n_local
                for cb in callbacks:
                    cb.on_train_batch_end(bi)
            tl, tc, tn = float(acc_loss), float(acc_correct), float(seen)
            if dp is not None and dp.active:
                tl, tc, tn = dp.allreduce_scalars([tl, tc, tn])
            logs = {"loss": tl / max(tn, 1.0) + float(self.l2_penalty()), "accuracy": tc / max(tn, 1.0)}
            if validation_data is not None:
                vl, va = self.evaluate(validation_data, dp=dp)
                logs["val_loss"], logs["val_accuracy"] = vl, va
            logs["learning_rate"] = float(opt["schedule"](self.opt_step) if "schedule" in opt
                                          else opt.get("lr", 0.0))
            for k, v in logs.items():
                hist.setdefault(k, []).append(float(v))
            if verbose and (dp is None or dp.rank == 0):
                log.info("Epoch %d/%d - %s", epoch + 1, epochs,
                         " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
            train_seq.on_epoch_end()
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end()

        class History:
            history = hist
        return History()

    # ----------------------------------------------------------- inference
    @torch.no_grad()
    def predict(self, x, batch_size: int = 256, verbose: Any = 0) -> np.ndarray:
        """probabilities [N,C] as a host array (keras Model.predict contract)."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        outs = []
        for b in range(0, x.shape[0], batch_size):
            xb = x[b:b + batch_size]
            outs.append(self.predict_device(xb).clone())
        return torch.cat(outs).cpu().numpy()

    def predict_device(self, x) -> torch.Tensor:
        x0 = self._input(x, False)
        if self.infer_dtype == "bf16" and self._bf16_storage_ok(x0.shape[2], x0.shape[3]):
            return self._forward_infer_bf16(x0)
        probs, _ = self.forward(x0, False)
        return probs

    def evaluate(self, data, verbose: Any = 0, dp=None) -> List[float]:
        """[loss, accuracy] over a sequence of (X, y) batches (one-hot or sparse labels); the
        loss uses the compiled label smoothing like keras' compiled loss.  With `dp`, sums are
        all-reduced so every rank returns the global result."""
        tot_loss = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device)
        count = 0
        prefetch = getattr(data, "prefetch", None)
        for i in range(len(data)):
            bx, by = data[i]
            if prefetch is not None and i + 1 < len(data):
                prefetch(i + 1)   # an uncached sequence decodes its next batch during this one
            if bx.shape[0] == 0:
                continue
            yt, idx = self._targets(by)
            probs, loss = self.forward(self._input(bx, False), False, yt)
            tot_loss += loss.sum()
            correct += (probs.argmax(-1) == idx).sum()
            count += int(idx.numel())
        tl, tc, tn = float(tot_loss), float(correct), float(count)
        if dp is not None and dp.active:
            tl, tc, tn = dp.allreduce_scalars([tl, tc, tn])
        return [tl / max(tn, 1.0) + float(self.l2_penalty()), tc / max(tn, 1.0)]

    # ------------------------------------------------------------- weights
    def weight_names(self) -> List[str]:
        names = []
        if self.norm is not None:
            names += ["input_norm.mean", "input_norm.variance"]
        bn_done = set()
        for name, _s, _k in self.specs:
            names.append(name)
            if name.endswith(".beta"):
                base = name[:-5]
                if base not in bn_done:
                    bn_done.add(base)
                    names += [base + ".moving_mean", base + ".moving_variance"]
        return names

    def get_weights(self) -> List[np.ndarray]:
        """All weights (trainable + BN moving statistics + normalization) as host arrays in
        keras layouts: conv kernels HWIO [k,k,Cin,Cout], 1x1 SE convs [1,1,in,out]."""
        out = []
        for name in self.weight_names():
            out.append(self._get_one(name))
        return out

    def _get_one(self, name: str) -> np.ndarray:
        if name == "input_norm.mean":
            return self.norm.mean.copy()
        if name == "input_norm.variance":
            return self.norm.variance.copy()
        if name.endswith(".moving_mean"):
            return self.s[name[:-12] + ".mean"].cpu().numpy()
        if name.endswith(".moving_variance"):
            return self.s[name[:-16] + ".var"].cpu().numpy()
        t = self.p[name].detach().cpu()
        kind = next(k for n_, _s, k in self.specs if n_ == name)
        if kind == "w3" or (kind == "w1" and t.dim() == 3):
            cin, taps, cout = t.shape
            k = int(round(math.sqrt(taps)))
            return t.view(cin, k, k, cout).permute(1, 2, 0, 3).contiguous().numpy()
        if kind == "w1":
            return t.view(1, 1, *t.shape).numpy().copy()
        return t.numpy().copy()

    def set_weights(self, weights: List[np.ndarray]) -> None:
        names = self.weight_names()
        if len(weights) != len(names):
            raise ValueError(f"set_weights: expected {len(names)} arrays, got {len(weights)}")
        for name, arr in zip(names, weights):
            arr = np.asarray(arr, dtype=np.float32)
            if name == "input_norm.mean":
                self.norm.mean = arr.reshape(3).copy()
            elif name == "input_norm.variance":
                self.norm.variance = arr.reshape(3).copy()
            elif name.endswith(".moving_mean"):
                self.s[name[:-12] + ".mean"].copy_(torch.from_numpy(arr))
            elif name.endswith(".moving_variance"):
                self.s[name[:-16] + ".var"].copy_(torch.from_numpy(arr))
            else:
                t = torch.from_numpy(arr)
                dst = self.p[name]
                if t.dim() == 4 and dst.dim() == 3:      # HWIO -> IKO
                    k = t.shape[0]
                    t = t.permute(2, 0, 1, 3).reshape(t.shape[2], k * k, t.shape[3])
                elif t.dim() == 4 and dst.dim() == 2:    # [1,1,in,out] -> [in,out]
                    t = t.reshape(t.shape[2], t.shape[3])
                dst.copy_(t.reshape(dst.shape))

    def ema_weights(self) -> List[np.ndarray]:
        """The EMA shadow in get_weights() order (train/utils.py:44-57 averages every weight)."""
        live_p, live_s = self.flat_p.clone(), self.flat_s.clone()
        self.flat_p.copy_(self.flat_ema)
        self.flat_s.copy_(self.flat_s_ema)
        w = self.get_weights()
        self.flat_p.copy_(live_p)
        self.flat_s.copy_(live_s)
        return w

    def config(self) -> Dict[str, Any]:
        return {"name": self.name, "num_classes": self.num_classes, "img_size": self.img_size,
                "use_norm": self.norm is not None, "widths": self.widths,
                "drop_block": self.drop_block, "drop_top": self.drop_top, "l2_reg": self.l2_reg,
                "separable": False, "augment": self.augment, "use_se": self.use_se}

    def save(self, path) -> None:
        """`leaf_cnn.keras`: a zip with config.json / metadata.json (keras-v3 member names) and
        model.weights.npz (authoritative here; HDF5 needs h5py, which this image lacks)."""
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        names = self.weight_names()
        arrays = {f"{i:03d}:{n}": a for i, (n, a) in enumerate(zip(names, self.get_weights()))}
        import io
        buf = io.BytesIO()
        np.savez(buf, **arrays)
        with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
            z.writestr("config.json", json.dumps({"module": "leaffliction_amd.model.cnn",
                                                   "class_name": "LeafCNN",
                                                   "config": self.config()}, indent=1))
            z.writestr("metadata.json", json.dumps({"leaffliction_amd_version": "0.1.0",
                                                     "weights_format": "npz",
                                                     "weight_names": names}, indent=1))
            z.writestr("model.weights.npz", buf.getvalue())


def load_model(path) -> LeafCNN:
    """Counterpart of keras.models.load_model for files written by LeafCNN.save.  A Keras-written
    `.keras` archive holds model.weights.h5 (HDF5; this image has no h5py) and is refused with a
    ValueError that says so — as is anything else that is not this package's npz archive."""
    import io
    path = Path(path)
    if not zipfile.is_zipfile(path):
        raise ValueError(f"{path}: not a .keras zip archive (legacy HDF5 / SavedModel files are not supported)")
    with zipfile.ZipFile(path) as z:
        members = set(z.namelist())
        if "model.weights.npz" not in members:
            if "model.weights.h5" in members:
                raise ValueError(f"{path}: Keras-written .keras archive (model.weights.h5, HDF5): not "
                                 "readable here — this backend stores model.weights.npz; retrain or "
                                 "convert with weight_names()/set_weights() (INTEGRATION.md)")
            raise ValueError(f"{path}: unsupported .keras archive: expected model.weights.npz or "
                             f"model.weights.h5, found {sorted(members)}")
        for need in ("config.json", "metadata.json"):
            if need not in members:
                raise ValueError(f"{path}: unsupported .keras archive: {need} is missing")
        cfg = json.loads(z.read("config.json"))["config"]
        data = np.load(io.BytesIO(z.read("model.weights.npz")))
        meta = json.loads(z.read("metadata.json"))
    if meta.get("weights_format") != "npz" or "weight_names" not in meta:
        raise ValueError(f"{path}: metadata.json does not describe a leaffliction_amd npz archive "
                         f"(weights_format={meta.get('weights_format')!r})")
    model = LeafCNN(num_classes=cfg["num_classes"], img_size=cfg["img_size"],
                    use_norm=cfg["use_norm"], widths=cfg["widths"], drop_block=cfg["drop_block"],
                    drop_top=cfg["drop_top"], l2_reg=cfg["l2_reg"], augment=cfg["augment"],
                    use_se=cfg["use_se"])
    keys = sorted(data.files)
    if [k.split(":", 1)[1] for k in keys] != meta["weight_names"]:
        raise ValueError(f"{path}: model.weights.npz does not hold the tensors metadata.json lists")
    model.set_weights([data[k] for k in keys])
    return model


def build_leafcnn(*, num_classes: int, img_size: int = 224, use_norm: bool = True,
                  widths: Optional[List[int]] = None, drop_block: float = 0.15,
                  drop_top: float = 0.40, l2_reg: float = 0.0, separable: bool = False,
                  augment: bool = True, use_se: bool = True, seed: int = 0):
    """Same keyword signature as the reference (cnn.py:52-64); returns (model, norm_layer)."""
    model = LeafCNN(num_classes=num_classes, img_size=img_size, use_norm=use_norm, widths=widths,
                    drop_block=drop_block, drop_top=drop_top, l2_reg=l2_reg, separable=separable,
                    augment=augment, use_se=use_se, seed=seed)
    return model, model.norm


def adapt_normalization(norm_layer, train_seq) -> None:
    """cnn.py:107-131: adapt on the first batches (<= 64 batches, >= 2048 samples)."""
    if norm_layer is None or not hasattr(norm_layer, "adapt"):
        return
    samples, collected = [], 0
    prefetch = getattr(train_seq, "prefetch", None)
    for i in range(min(len(train_seq), 64)):
        batch = train_seq[i]
        if prefetch is not None and i + 1 < min(len(train_seq), 64):
            prefetch(i + 1)
        X = batch[0] if isinstance(batch, (list, tuple)) else batch
        if isinstance(X, torch.Tensor):
            X = X.cpu().numpy()
        if X.dtype == np.uint8:
            X = X.astype(np.float32) / 255.0
        samples.append(X)
        collected += len(X)
        if collected >= 2048:
            break
    if samples:
        norm_layer.adapt(np.concatenate(samples, axis=0))
