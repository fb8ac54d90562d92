"""End-to-end `model.fit` throughput on a device-resident synthetic dataset (MI355X only):
the real training loop (ManifestSequence batches, label upload, callbacks) rather than the
bare step bench.py times.  usage: python scripts/fit_throughput.py [n_images] [batch]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, ".")
from leaffliction_amd.dataio.manifest import ManifestItem  # noqa: E402
from leaffliction_amd.dataio.sequence import ManifestSequence  # noqa: E402
from leaffliction_amd.model.cnn import build_leafcnn  # noqa: E402
from leaffliction_amd.train.utils import CosineDecay, build_loss, build_optimizer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rng = np.random.RandomState(0)
    with tempfile.TemporaryDirectory() as d:
        items, labels = [], [f"Plant__c{k}" for k in range(8)]
        for i in range(64):  # 64 distinct files, reused: decode cost is not what is measured
            p = Path(d) / f"img{i}.jpg"
            Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8)).save(p, quality=90)
        for i in range(n):
            lab = labels[i % 8]
            items.append(ManifestItem(id=str(i), plant="Plant", cls=lab.split("__")[1], label=lab,
                                      split="train", src=Path(d) / f"img{i % 64}.jpg"))
        l2i = {lab: k for k, lab in enumerate(labels)}
        t0 = time.perf_counter()
        seq = ManifestSequence(items, l2i, 224, bs, True, 42, num_classes=8, one_hot=True, cache=True, workers=8)
        print(f"cache build: {time.perf_counter() - t0:.1f} s for {n} images "
              f"({seq._cache_dev.numel() / 1e9:.2f} GB in HBM)")
        model, _ = build_leafcnn(num_classes=8, img_size=224, seed=1)
        cfg = {"optimizer": "adamw", "lr": 2e-3, "weight_decay": 1e-4, "label_smoothing": 0.02,
               "cosine_decay": True, "ema_decay": 0.999, "clipnorm": 0.5}
        model.compile(build_optimizer(cfg, CosineDecay(2e-3, len(seq) * 3)), build_loss(cfg), ["accuracy"])
        model.fit(seq, epochs=1, verbose=0)  # warm-up epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit(seq, epochs=2, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"fit: {2 * n / dt:.0f} images/s ({dt / (2 * len(seq)) * 1e3:.1f} ms per {bs}-image step)")


if __name__ == "__main__":
    main()
