"""Augmentation-pass table of bench.py on its own (development aid):
python scripts/bench_augment.py [n_images] [iters]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    out = bench.augment_throughput(torch.device("cuda:0"), n=n, iters=iters)
    for k, v in out.items():
        print(k, json.dumps(v))
