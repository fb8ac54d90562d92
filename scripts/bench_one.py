"""Time single augmentation cases of bench.py (development aid):
python scripts/bench_one.py flip,pack [n_images] [iters]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    names = sys.argv[1].split(",")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    out = bench.augment_throughput(torch.device("cuda:0"), n=n, iters=iters, only=names)
    print(os.environ.get("LEAFHIP_LIB", "default"), {k: v["GB_s"] for k, v in out.items() if isinstance(v, dict)})
