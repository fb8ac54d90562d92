"""Time the bf16 weight-gradient kernel on one layer shape (python scripts/microbench/wgrad_modes.py [cin cout hw])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leaffliction_amd import nn  # noqa: E402


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    hw = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    n, dev, bf = 256, torch.device("cuda:0"), torch.bfloat16
    torch.manual_seed(0)
    x = torch.randn(n, cin, hw, hw, device=dev).to(bf)
    g = torch.randn(n, cout, hw, hw, device=dev).to(bf)
    dw = torch.zeros(cin, 9, cout, device=dev)
    for _ in range(2):
        nn.conv2d_wgrad_bf16(x, g, 3, out=dw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        nn.conv2d_wgrad_bf16(x, g, 3, out=dw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"wgrad {cin}->{cout}@{hw}: {ms * 1e3:.1f} us, {(x.numel() + g.numel()) * 2 / ms / 1e9:.2f} TB/s, checksum {float(dw.sum()):.4f}")


if __name__ == "__main__":
    main()
