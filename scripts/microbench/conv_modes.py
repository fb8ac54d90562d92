"""Time the 32->32 3x3 @224 bf16 convolution in each mode the training step uses it in
(python scripts/microbench/conv_modes.py [cin cout hw])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leaffliction_amd import nn  # noqa: E402


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    hw = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    hh = int(sys.argv[4]) if len(sys.argv) > 4 else hw     # height (plane stride = hh * hw elements)
    n, dev = 256, torch.device("cuda:0")
    bf = torch.bfloat16
    x = torch.randn(n, cin, hh, hw, device=dev).to(bf)
    y = torch.randn(n, cout, hh, hw, device=dev).to(bf)
    out = torch.zeros(n, cout, hh, hw, device=dev, dtype=bf)
    wp = nn.conv2d_bf16_weights(torch.randn(cin, 9, cout, device=dev) * 0.05, 3)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    msc, msh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    piv = torch.zeros(cout, device=dev)
    modes = {
        "plain": dict(),
        "fwd: prologue + stats": dict(in_scale=sc, in_shift=sh, in_relu=True, stats=True, pivot=piv),
        "fwd: stats only": dict(stats=True, pivot=piv),
        "dgrad: accumulate": dict(accumulate=True),
        "dgrad: mask + sums": dict(mask_y=y, mask_scale=msc, mask_shift=msh, mask_relu=True),
        "dgrad: accumulate + mask + sums": dict(accumulate=True, mask_y=y, mask_scale=msc, mask_shift=msh, mask_relu=True),
    }
    for name, kw in modes.items():
        for _ in range(2):
            nn.conv2d_bf16_train(x, wp, cout, 3, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            nn.conv2d_bf16_train(x, wp, cout, 3, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        base = (x.numel() + out.numel()) * 2
        extra = (out.numel() * 2 if kw.get("accumulate") else 0) + (y.numel() * 2 if "mask_y" in kw else 0)
        print(f"{name:34s} {ms * 1e3:8.1f} us   {(base + extra) / ms / 1e9:7.2f} TB/s algorithmic   {ms * 1e6 / (n * hh * hw):.4f} ns/pixel")


if __name__ == "__main__":
    main()
