"""A few launches of the geometric augmentation kernels for PMC collection (rocprofv3 --pmc ...)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from leaffliction_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, S = 2048, 224
g = torch.Generator().manual_seed(42)
x = torch.randint(0, 256, (n, S, S, 3), dtype=torch.uint8, generator=g).to(dev)
rng = np.random.RandomState(42)
f = rng.uniform(0.05, 0.15, n)
skew = torch.tensor([[1 + v, 0, -v * S, 0, 1 + v, -v * S, 0, 0] for v in f], dtype=torch.float64, device=dev)
sh = rng.uniform(-0.2, 0.2, n)
shear = torch.tensor([[1, v, 0, 0, 1, 0, 0, 0] if i % 2 else [1, 0, 0, v, 1, 0, 0, 0] for i, v in enumerate(sh)],
                     dtype=torch.float64, device=dev)
rplan = ops.rotate_expand_plan(S, S, rng.uniform(-30, 30, n), dev)
rbuf = torch.empty(rplan["total"], dtype=torch.uint8, device=dev)
boxes = []
for _ in range(n):
    r = rng.uniform(0.8, 0.95)
    nw = nh = int(S * r)
    boxes.append((rng.randint(0, S - nw + 1), rng.randint(0, S - nh + 1), nw, nh))
ctab = ops.crop_resize_plan(S, S, boxes, dev)
for _ in range(2):
    ops.warp_bicubic_u8(x, skew, True, True)
    ops.warp_bicubic_u8(x, shear, False)
    ops.rotate_expand_apply(x, rplan, 255, rbuf)
    ops.resample_u8(x, S, S, ctab[0], ctab[1], ctab[2], ctab[3], True, ctab[4])
torch.cuda.synchronize()
