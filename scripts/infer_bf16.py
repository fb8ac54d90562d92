"""A few batch-1024 forward passes in the bf16 inference mode (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaffliction_amd.model.cnn import LeafCNN  # noqa: E402

dev = torch.device("cuda:0")
m = LeafCNN(num_classes=8, img_size=224, widths=[32, 64, 128, 256], drop_block=0.15, drop_top=0.4,
            l2_reg=1e-4, augment=True, use_se=True, seed=42, device=dev)
m.set_inference_dtype(sys.argv[1] if len(sys.argv) > 1 else "bf16")
x = torch.randint(0, 256, (1024, 224, 224, 3), dtype=torch.uint8, device=dev)
for _ in range(4):
    m.predict_device(x)
torch.cuda.synchronize()
