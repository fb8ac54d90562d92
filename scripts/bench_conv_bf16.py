"""Per-layer time of the bf16-operand forward convolution at the leaf_cnn shapes (batch 1024)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaffliction_amd import nn  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SHAPES = [(3, 32, 224, 3), (32, 32, 224, 3), (32, 64, 112, 3), (64, 64, 112, 3), (32, 64, 112, 1), (64, 128, 56, 3),
          (128, 128, 56, 3), (64, 128, 56, 1), (128, 256, 28, 3), (256, 256, 28, 3), (128, 256, 28, 1)]
for cin, cout, s, k in SHAPES:
    x = torch.randn((n, cin, s, s), device=dev)
    w = torch.randn((cin, k * k, cout), device=dev) * 0.05
    wp = nn.conv2d_bf16_weights(w, k)
    out = torch.empty((n, cout, s, s), device=dev)
    nn.conv2d_bf16(x, wp, cout, k, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        nn.conv2d_bf16(x, wp, cout, k, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flop = 2.0 * n * s * s * cin * cout * k * k
    byts = 4.0 * n * s * s * (cin + cout)
    print(f"{cin:4d}->{cout:4d} {s:4d}^2 k{k}: {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TF/s  {byts / ms / 1e9:6.2f} TB/s", flush=True)
