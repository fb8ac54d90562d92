"""Where Predictor.predict_batch (pooled path) spends the main thread's time (cProfile; development aid)."""
import cProfile
import json
import os
import pstats
import shutil
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    from leaffliction_amd.model.cnn import LeafCNN
    from leaffliction_amd.predict.predictor import Predictor
    dev = torch.device("cuda:0")
    tmp = Path(tempfile.mkdtemp(prefix="lf_ppred_"))
    try:
        src = tmp / "images"
        bench._e2e_make_dataset(src, dev, bench.usable_cores(), bench._e2e_layout(11000))
        files = sorted(str(p) for p in src.rglob("*.JPG"))[:4096]
        os.chdir(tmp)
        model = LeafCNN(num_classes=bench.NUM_CLASSES, img_size=bench.IMG, widths=bench.WIDTHS, drop_block=0.15, drop_top=0.40,
                        l2_reg=1e-4, augment=True, use_se=True, seed=42, device=dev)
        model.norm.mean[:] = 0.5
        model.norm.variance[:] = 1.0 / 12.0
        learn = tmp / "artifacts" / "models"
        learn.mkdir(parents=True)
        model.save(str(learn / "leaf_cnn.keras"))
        (learn / "meta.json").write_text(json.dumps({"model_file": str(learn / "leaf_cnn.keras"),
                                                     "labels": [f"class_{i}" for i in range(bench.NUM_CLASSES)],
                                                     "data": {"img_size": bench.IMG}}))
        os.environ["LEAFFLICTION_INFER_DTYPE"] = "bf16"
        pred = Predictor(learn)
        pred.load()
        pred.predict_batch(files[:256])
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        pred.predict_batch(files)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(25)
        st.sort_stats("tottime").print_stats(25)
        pred.close()
    finally:
        os.chdir("/")
        shutil.rmtree(tmp, ignore_errors=True)
