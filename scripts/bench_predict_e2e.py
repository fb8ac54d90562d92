"""bench.py's end-to-end predict line on its own (development aid): python scripts/bench_predict_e2e.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.predict_end_to_end(torch.device("cuda:0"))))
