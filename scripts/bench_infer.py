"""Forward-only rates of bench.py on their own (development aid): python scripts/bench_infer.py [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    from leaffliction_amd import _lib
    from leaffliction_amd.model.cnn import LeafCNN
    _lib.load()
    model = LeafCNN(num_classes=bench.NUM_CLASSES, img_size=bench.IMG, widths=bench.WIDTHS, drop_block=0.15,
                    drop_top=0.40, l2_reg=1e-4, augment=True, use_se=True, seed=42, device=dev)
    model.norm.mean[:] = 0.5
    model.norm.variance[:] = 1.0 / 12.0
    print(json.dumps(bench.inference_throughput(model, dev, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 1024)))
