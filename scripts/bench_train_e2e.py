"""Training from JPEG files, uncached (the reference's base preset: cache=False, srcs/cli/train.py:38), one epoch
of `fit`: the loader with the next batch decoding on the codec workers against the plain host loader.
Development aid: python scripts/bench_train_e2e.py [files] [batch]"""
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from leaffliction_amd.dataio.manifest import ManifestItem  # noqa: E402
from leaffliction_amd.dataio.sequence import ManifestSequence  # noqa: E402
from leaffliction_amd.model.cnn import build_leafcnn  # noqa: E402
from leaffliction_amd.train.utils import build_loss, build_optimizer  # noqa: E402


def epoch(items, l2i, batch, pooled):
    seq = ManifestSequence(items, l2i, bench.IMG, batch, True, 42, num_classes=len(l2i), one_hot=True)
    if not pooled:
        seq.prefetch = lambda idx: None
        seq.POOL_MIN = 10 ** 9
    cfg = {"optimizer": "adamw", "lr": 1e-3, "weight_decay": 1e-4, "label_smoothing": 0.02,
           "cosine_decay": False, "ema_decay": 0.0, "clipnorm": 0.5}
    model, _ = build_leafcnn(num_classes=len(l2i), img_size=bench.IMG, widths=list(bench.WIDTHS), drop_block=0.15,
                             drop_top=0.40, l2_reg=1e-4, seed=42)
    model.compile(build_optimizer(cfg, 1e-3), build_loss(cfg), ["accuracy"])
    warm = ManifestSequence(items[:4 * batch], l2i, bench.IMG, batch, False, 42, num_classes=len(l2i), one_hot=True)
    if not pooled:
        warm.prefetch = lambda idx: None
        warm.POOL_MIN = 10 ** 9
    model.fit(warm, epochs=1, verbose=0)        # graph capture, worker start-up
    seq._decoder, warm._decoder = warm._decoder, None   # the codec workers live across epochs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit(seq, epochs=1, verbose=0)
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    seq.close()
    return len(items) / sec


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    tmp = Path(tempfile.mkdtemp(prefix="lf_train_"))
    try:
        bench._e2e_make_dataset(tmp / "images", dev, bench.usable_cores(), bench._e2e_layout(11000))
        files = sorted((tmp / "images").rglob("*.JPG"))[:n_files]
        labels = sorted({f.parent.name for f in files})
        l2i = {la: i for i, la in enumerate(labels)}
        items = [ManifestItem(str(i), "p", f.parent.name, f.parent.name, "train", f) for i, f in enumerate(files)]
        out = {"files": len(items), "batch": batch, "dtype": os.environ.get("LEAFFLICTION_TRAIN_DTYPE", "f32") + " step",
               "images_per_sec_prefetching_loader": round(epoch(items, l2i, batch, True), 1),
               "images_per_sec_host_loader": round(epoch(items, l2i, batch, False), 1)}
        print(json.dumps(out))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
