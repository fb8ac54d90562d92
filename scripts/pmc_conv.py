"""Run a few conv launches for PMC collection (rocprofv3 --pmc ...)."""
import sys
import torch
sys.path.insert(0, ".")
from leaffliction_amd import nn
dev = torch.device("cuda:0")
n = 256
for cin, cout, hw in [(32, 32, 224), (64, 64, 112), (128, 128, 56), (256, 256, 28)]:
    x = torch.randn(n, cin, hw, hw, device=dev)
    w = torch.randn(cin, 9, cout, device=dev) * 0.05
    dy = torch.randn(n, cout, hw, hw, device=dev)
    y = torch.empty(n, cout, hw, hw, device=dev)
    for _ in range(2):
        nn.conv2d(x, w, 3, out=y)
        nn.conv2d_wgrad(x, dy, 3)
    torch.cuda.synchronize()
    del x, w, dy, y
