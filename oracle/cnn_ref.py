"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement (torch fp32 on the host, autograd for the backward pass) of the reference's
`leaf_cnn` training step: srcs/model/cnn.py:9-104 (topology), srcs/train/utils.py:17-57
(AdamW + per-variable clipnorm, label-smoothed CCE, EMA), srcs/cli/train.py:30-50,266-329
(presets, cosine schedule).  The arithmetic lives in Keras 3 / TensorFlow (requirements.txt:2-3,
`keras>=3.0.0`, `tensorflow>=2.15`, unpinned), which are not installed here and cannot be
(no network); the reference has no tests or golden vectors for this path.

PARITY UNPINNED against Keras bits: the layer semantics below restate the Keras 3
documentation/source and are guarded by known-answer tests
(tests/test_oracle_kats.py).  The HIP path is compared against THIS file at the float
tolerances stated in tests/test_cnn_gpu.py.

Parameter layouts are the product's (conv kernels "IKO" [Cin, k*k, Cout]; dense [F, C]) so
weights and gradients compare tensor-for-tensor.  Stochastic inputs (augmentation draws,
dropout masks) are explicit arguments.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.99
BN_EPS = 1e-3
NORM_EPS = 1e-7  # keras.backend.epsilon()


def preset(scale: str):
    """train.py:266-280."""
    if scale == "tiny":
        return [16, 32, 64], 0.10, 0.30
    if scale == "small":
        return [32, 64, 128], 0.15, 0.35
    return [32, 64, 128, 256], 0.15, 0.40


def param_specs(num_classes: int, widths: List[int]):
    """Ordered (name, shape, kind, l2) — kind: 'w3' 3x3 kernel, 'w1' 1x1, 'vec', 'dense'."""
    specs = [("stem.w", (3, 9, widths[0]), "w3"), ("stem.bn.gamma", (widths[0],), "vec"),
             ("stem.bn.beta", (widths[0],), "vec")]
    cin = widths[0]
    for i, f in enumerate(widths):
        p = f"s{i}."
        specs += [(p + "c1.w", (cin, 9, f), "w3"), (p + "bn1.gamma", (f,), "vec"),
                  (p + "bn1.beta", (f,), "vec"),
                  (p + "c2.w", (f, 9, f), "w3"), (p + "bn2.gamma", (f,), "vec"),
                  (p + "bn2.beta", (f,), "vec"),
                  (p + "se.w1", (f, f // 8), "w1"), (p + "se.b1", (f // 8,), "vec"),
                  (p + "se.w2", (f // 8, f), "w1"), (p + "se.b2", (f,), "vec")]
        if cin != f:
            specs += [(p + "proj.w", (cin, 1, f), "w1"), (p + "bnp.gamma", (f,), "vec"),
                      (p + "bnp.beta", (f,), "vec")]
        cin = f
    specs += [("dense.w", (widths[-1], num_classes), "dense"), ("dense.b", (num_classes,), "vec")]
    return specs


def bn_names(widths: List[int]):
    names = ["stem.bn"]
    cin = widths[0]
    for i, f in enumerate(widths):
        names += [f"s{i}.bn1", f"s{i}.bn2"]
        if cin != f:
            names.append(f"s{i}.bnp")
        cin = f
    return names


def init_params(num_classes: int, widths: List[int], seed: int = 0) -> Dict[str, torch.Tensor]:
    """glorot_uniform kernels, zero biases, BN gamma=1 beta=0 (keras defaults)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind in param_specs(num_classes, widths):
        if kind in ("w3", "w1") and len(shape) == 3:
            fan_in, fan_out = shape[0] * shape[1], shape[2] * shape[1]
        elif kind in ("w1", "dense"):
            fan_in, fan_out = shape[0], shape[1]
        if kind == "vec":
            out[name] = torch.ones(shape) if name.endswith("gamma") else torch.zeros(shape)
        else:
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            out[name] = (torch.rand(shape, generator=g) * 2 - 1) * lim
    return out


def init_state(widths: List[int]) -> Dict[str, torch.Tensor]:
    st = {}
    cin = widths[0]
    chans = {"stem.bn": widths[0]}
    for i, f in enumerate(widths):
        chans[f"s{i}.bn1"] = f
        chans[f"s{i}.bn2"] = f
        if cin != f:
            chans[f"s{i}.bnp"] = f
        cin = f
    for k, c in chans.items():
        st[k + ".mean"] = torch.zeros(c)
        st[k + ".var"] = torch.ones(c)
    return st


def conv(x, w_iko, k):
    cin, taps, cout = w_iko.shape
    w = w_iko.permute(2, 0, 1).reshape(cout, cin, k, k)
    return F.conv2d(x, w, padding=k // 2)


def batchnorm(y, gamma, beta, state, key, training):
    """keras BatchNormalization(axis=-1): batch mean / biased var in training (and the moving
    statistics are updated in place), moving statistics at inference."""
    if training:
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            state[key + ".mean"].mul_(BN_MOMENTUM).add_(mean.detach() * (1 - BN_MOMENTUM))
            state[key + ".var"].mul_(BN_MOMENTUM).add_(var.detach() * (1 - BN_MOMENTUM))
    else:
        mean, var = state[key + ".mean"], state[key + ".var"]
    inv = torch.rsqrt(var + BN_EPS)
    return (y - mean.view(1, -1, 1, 1)) * (inv * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def input_stage(x_u8: torch.Tensor, aug: Optional[torch.Tensor], mean=None, denom=None):
    """[N,H,W,3] u8 -> [N,3,H,W] f32: /255, RandomFlip -> RandomRotation(bilinear, reflect) ->
    RandomContrast (cnn.py:74-83), Normalization (cnn.py:84-86).  aug [N,4] = flip,cos,sin,contrast."""
    x = x_u8.float()
    n, h, w, _ = x.shape
    if aug is not None:
        out = torch.empty_like(x)
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                                torch.arange(w, dtype=torch.float32), indexing="ij")
        for i in range(n):
            flip, cs, sn, ct = [float(v) for v in aug[i]]
            cs, sn = torch.tensor(cs), torch.tensor(sn)
            wm, hm = float(w - 1), float(h - 1)
            xoff = (wm - (cs * wm - sn * hm)) * 0.5
            yoff = (hm - (sn * wm + cs * hm)) * 0.5
            fx = cs * xs - sn * ys + xoff
            fy = sn * xs + cs * ys + yoff
            x0, y0 = torch.floor(fx), torch.floor(fy)
            ax, ay = (fx - x0).unsqueeze(-1), (fy - y0).unsqueeze(-1)

            def refl(v, size):
                v = torch.remainder(v.long(), 2 * size)
                return torch.where(v < size, v, 2 * size - 1 - v)

            xa, xb = refl(x0, w), refl(x0 + 1, w)
            ya, yb = refl(y0, h), refl(y0 + 1, h)
            if flip:
                xa, xb = w - 1 - xa, w - 1 - xb
            img = x[i]
            top = img[ya, xa] + (img[ya, xb] - img[ya, xa]) * ax
            bot = img[yb, xa] + (img[yb, xb] - img[yb, xa]) * ax
            v = (top + (bot - top) * ay) * (1.0 / 255.0)
            mu = v.mean(dim=(0, 1), keepdim=True)
            out[i] = torch.clamp((v - mu) * ct + mu, 0.0, 255.0)
        x = out
    else:
        x = x / 255.0
    if mean is not None:
        x = (x - torch.tensor(mean, dtype=torch.float32)) / torch.tensor(denom, dtype=torch.float32)
    return x.permute(0, 3, 1, 2).contiguous()


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest-even bf16 -> fp32 (what a bf16 store followed by a load yields).  A float64 tensor comes back
    as float64 (the rounding is the same; tests run the restatement in double precision to see how far a change of
    ACCUMULATION precision alone moves a bf16 step)."""
    return t.to(torch.bfloat16).to(t.dtype)


def _q(t: torch.Tensor, round_grad: bool = True) -> torch.Tensor:
    """Value rounded to bf16 in the forward pass (straight-through); with round_grad the gradient
    that arrives at the rounded tensor is rounded to bf16 too (a bf16 gradient store)."""
    out = t + (bf16_round(t) - t).detach()
    if round_grad and out.requires_grad:
        out.register_hook(bf16_round)
    return out


def forward(params, state, x, widths, training, drops=None, top_drop=None, collect=None, lowp=False):
    """x [N,3,H,W] f32 (already normalised) -> probabilities [N,C].

    drops[i] [N,C_i] SpatialDropout2D keep-scales (0 or 1/(1-p)); top_drop [N,F] likewise.

    lowp=True restates the MIXED-PRECISION step (the reference's default policy is Keras'
    mixed_float16, srcs/cli/train.py:179-190; BASELINE configs[3] names bf16): 16-bit storage of
    every layer output and gradient, 16-bit matrix operands, fp32 variables / BatchNorm statistics
    / SE / softmax / loss.  The rounding points are the ones the HIP step has (include/leafhip.h,
    "mixed-precision TRAINING step"): convolution operands (activation and kernel) and outputs,
    pooled block outputs; in the backward pass the gradient w.r.t. every convolution output and
    input, the pooled gradient after the dropout factor, and the head's gradient w.r.t. the last
    pooled map.  BatchNorm statistics are those of the rounded convolution output.
    """
    P = params
    rnd = (lambda t: _q(t)) if lowp else (lambda t: t)           # value + its gradient
    opd = (lambda t: _q(t, False)) if lowp else (lambda t: t)    # operand rounding only
    wq = {k: opd(v) for k, v in P.items() if k.endswith(".w") and v.dim() == 3} if lowp else P

    def cbn(inp, wname, k, bn):
        y = rnd(conv(inp, wq[wname], k))                         # stored conv output; dY rounded
        return batchnorm(y, P[bn + ".gamma"], P[bn + ".beta"], state, bn, training)

    a = torch.relu(cbn(opd(x), "stem.w", 3, "stem.bn"))
    cin = widths[0]
    for i, f in enumerate(widths):
        p = f"s{i}."
        if lowp and a.requires_grad:
            a.register_hook(bf16_round)        # the block's input gradient (all consumers summed)
        sc = a
        y = torch.relu(cbn(rnd(a) if cin != f else opd(a), p + "c1.w", 3, p + "bn1"))
        y = torch.relu(cbn(rnd(y), p + "c2.w", 3, p + "bn2"))
        m = y.mean(dim=(2, 3))
        z = torch.relu(m @ P[p + "se.w1"] + P[p + "se.b1"])
        s = torch.sigmoid(z @ P[p + "se.w2"] + P[p + "se.b2"])
        y = y * s.view(s.shape[0], s.shape[1], 1, 1)
        if cin != f:
            sc = cbn(rnd(sc), p + "proj.w", 1, p + "bnp")
        r = torch.relu(sc + y)
        mp = F.max_pool2d(r, 2)
        if lowp and mp.requires_grad:
            mp.register_hook(bf16_round)       # dr = bf16(dp * drop), then routed
        if training and drops is not None:
            mp = mp * drops[i].view(r.shape[0], r.shape[1], 1, 1)
        a = opd(mp)                            # stored pooled output
        if collect is not None:
            collect[p + "out"] = a
        cin = f
    if lowp and a.requires_grad:
        a.register_hook(bf16_round)            # head gradient spread over the last pooled map
    g = a.mean(dim=(2, 3))
    if training and top_drop is not None:
        g = g * top_drop
    logits = g @ P["dense.w"] + P["dense.b"]
    return torch.softmax(logits, dim=-1)


def cce_loss(probs, y_true):
    """keras categorical_crossentropy on probabilities: renormalise, clip [1e-7, 1-1e-7]."""
    p = probs / probs.sum(dim=-1, keepdim=True)
    p = torch.clamp(p, 1e-7, 1 - 1e-7)
    return -(y_true * torch.log(p)).sum(dim=-1)


def smooth_labels(y_onehot, smoothing):
    c = y_onehot.shape[-1]
    return y_onehot * (1.0 - smoothing) + smoothing / c


def l2_penalty(params, widths, num_classes, l2):
    tot = 0.0
    for name, _shape, kind in param_specs(num_classes, widths):
        if kind == "w3":
            tot = tot + l2 * (params[name] ** 2).sum()
    return tot


def cosine_lr(lr0, step, total):
    """keras CosineDecay(alpha=0) at 0-based `step` (train.py:313-318)."""
    s = min(step, total)
    return lr0 * 0.5 * (1.0 + math.cos(math.pi * s / total))


def adamw_step(params, grads, m, v, step, lr, wd=1e-4, clipnorm=0.5, b1=0.9, b2=0.999, eps=1e-7):
    """Keras 3 AdamW.update: clip_by_norm per variable, decoupled decay, Adam (1-based step)."""
    alpha = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    for k in params:
        g = grads[k]
        if clipnorm and clipnorm > 0:
            norm = torch.sqrt((g.double() ** 2).sum()).float()
            g = g * clipnorm / torch.maximum(norm, torch.tensor(clipnorm))
        w = params[k]
        w = w - w * wd * lr
        m[k] = m[k] + (g - m[k]) * (1 - b1)
        v[k] = v[k] + (g * g - v[k]) * (1 - b2)
        params[k] = w - m[k] * alpha / (torch.sqrt(v[k]) + eps)
    return params, m, v


def train_step(params, state, x, y_onehot, widths, drops, top_drop, l2=1e-4, smoothing=0.02,
               grads_include_l2=True, lowp=False):
    """One forward/backward: returns (loss incl. L2, data_loss, probs, grads dict).

    With grads_include_l2=False the gradients are those of the data loss alone (the HIP path
    adds the regulariser's 2*l2*w inside its optimizer kernel).  lowp: see forward()."""
    leaf = {k: t.clone().requires_grad_(True) for k, t in params.items()}
    probs = forward(leaf, state, x, widths, True, drops, top_drop, lowp=lowp)
    yt = smooth_labels(y_onehot, smoothing) if smoothing > 0 else y_onehot
    data_loss = cce_loss(probs, yt).mean()
    loss = data_loss + l2_penalty(leaf, widths, y_onehot.shape[-1], l2)
    (loss if grads_include_l2 else data_loss).backward()
    grads = {k: t.grad.detach().clone() for k, t in leaf.items()}
    return float(loss.detach()), float(data_loss.detach()), probs.detach(), grads
