"""Forward-only rate of the fp32 and bf16 inference modes at batch 1024 (bench.py's inference line alone)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaffliction_amd.model.cnn import LeafCNN  # noqa: E402

dev = torch.device("cuda:0")
m = LeafCNN(num_classes=8, img_size=224, widths=[32, 64, 128, 256], drop_block=0.15, drop_top=0.4,
            l2_reg=1e-4, augment=True, use_se=True, seed=42, device=dev)
x = torch.randint(0, 256, (1024, 224, 224, 3), dtype=torch.uint8, device=dev)
for rep in range(2):
    for mode in ("f32", "bf16"):
        m.set_inference_dtype(mode)
        m.predict_device(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.predict_device(x)
        torch.cuda.synchronize()
        print(mode, round(1024 * 5 / (time.perf_counter() - t0)), "img/s", flush=True)
