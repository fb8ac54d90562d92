"""One augmentation op on a resident batch, a few launches: the target of rocprofv3 runs
(python scripts/prof_one_op.py blur15|blur5|hist|<row of bench.py's augmentation table> [n_images])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaffliction_amd import ops  # noqa: E402

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "blur15"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
    for _ in range(3):
        if what == "blur15":
            ops.gauss_blur_u8(x, 15, 0.0)
        elif what == "blur5":
            ops.gauss_blur_u8(x, 5, 0.0)
        elif what == "hist":
            ops.hist_u8(x)
        else:
            break
    torch.cuda.synchronize()
    if what not in ("blur15", "blur5", "hist"):   # any row of bench.py's augmentation table
        import bench
        print(bench.augment_throughput(dev, n=n, iters=3, only=[what]))
