"""bf16 inference vs fp32 inference on the same model at several batch sizes (max |dp|, label agreement)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaffliction_amd.model.cnn import LeafCNN  # noqa: E402

dev = torch.device("cuda:0")
m = LeafCNN(num_classes=8, img_size=224, widths=[32, 64, 128, 256], l2_reg=1e-4, seed=42, device=dev)
g = torch.Generator().manual_seed(1)
for n in (8, 256, 1024):
    x = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
    m.set_inference_dtype("f32")
    p32 = m.predict_device(x).clone()
    m.set_inference_dtype("bf16")
    p16 = m.predict_device(x).clone()
    torch.cuda.synchronize()
    d = (p16 - p32).abs()
    print(n, "max|dp|", float(d.max()), "nan", bool(torch.isnan(p16).any()), "labels equal",
          float((p16.argmax(-1) == p32.argmax(-1)).float().mean()), "worst image", int(d.max(-1).values.argmax()))
# images 512.. = copies of images 0..511: every per-image result must repeat bit for bit
x = torch.randint(0, 256, (512, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
x2 = torch.cat([x, x])
for mode in ("f32", "bf16"):
    m.set_inference_dtype(mode)
    p = m.predict_device(x2).clone()
    torch.cuda.synchronize()
    print(mode, "halves equal:", bool(torch.equal(p[:512], p[512:])), "max diff", float((p[:512] - p[512:]).abs().max()))
m.set_inference_dtype("f32")
p32 = m.predict_device(x2).clone()
m.set_inference_dtype("bf16")
p16 = m.predict_device(x2).clone()
eq = (p16.argmax(-1) == p32.argmax(-1)).float()
print("label agreement first half", float(eq[:512].mean()), "second half", float(eq[512:].mean()),
      "top-2 margin median (fp32)", float((p32.topk(2).values[:, 0] - p32.topk(2).values[:, 1]).median()))
