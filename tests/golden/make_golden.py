"""Generate golden vectors by running the REFERENCE's own code in this container.

Run once here (the reference cannot travel to the GPU box):
    python tests/golden/make_golden.py
It imports `/root/reference/srcs/...` (the PIL/numpy half imports cleanly: SURVEY §8c),
feeds it seeded synthetic JPEGs and stores inputs (decoded arrays), the parameters the
reference drew from its RNGs, and outputs — both the PIL image the reference handed to
`ImageLoader.save_pil_image` (intercepted at that boundary, pre-JPEG) and nothing else of the
reference.  Only data is written: `tests/golden/*.npz|*.json`.
"""
from __future__ import annotations

import json
import random
import sys
import tempfile
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REF))

from srcs.preprocessing.image_augmenter import ImageAugmenter  # noqa: E402
from srcs.preprocessing.dataset_components import AugmentationPlanner  # noqa: E402
from srcs.utils import image_utils  # noqa: E402
from srcs.utils.image_utils import ImageLoader, ImageTransforms  # noqa: E402
from srcs.dataio.manifest import ManifestItem, build_label_mapping  # noqa: E402
from srcs.utils.confusion_matrix import compute_confusion_counts  # noqa: E402


def leaf_like(h: int, w: int, seed: int) -> np.ndarray:
    """SURVEY §8d set L recipe scaled to (h, w)."""
    rng = np.random.RandomState(seed)
    img = np.clip(rng.normal(150, 8, (h, w, 1)).repeat(3, axis=2), 0, 255)
    yy, xx = np.mgrid[0:h, 0:w]
    s = min(h, w) / 224.0
    cy, cx = rng.randint(int(80 * s), int(143 * s) + 1, 2)
    r = rng.randint(int(50 * s), int(89 * s) + 1)
    disc = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    img[disc] = (60, 140, 50)
    for _ in range(rng.randint(0, 6)):
        by, bx = rng.randint(0, h), rng.randint(0, w)
        br = rng.randint(max(1, int(3 * s)), max(2, int(10 * s)) + 1)
        img[(yy - by) ** 2 + (xx - bx) ** 2 <= br * br] = (120, 70, 30)
    img = img + rng.normal(0, 8, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main() -> None:
    captured = {}
    orig_save = ImageLoader.save_pil_image

    def capture(img, output_path, quality=95):
        captured["img"] = np.array(img)
        orig_save(img, output_path, quality)

    ImageLoader.save_pil_image = staticmethod(capture)
    image_utils.ImageLoader.save_pil_image = staticmethod(capture)

    sizes = [(48, 64), (96, 96), (224, 224)]
    ops = ["flip", "rotate", "skew", "shear", "crop", "distortion"]
    seeds = [1, 7, 123456]
    arrays = {}
    cases = []
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        for si, (h, w) in enumerate(sizes):
            src = td / f"in_{si}.jpg"
            Image.fromarray(leaf_like(h, w, 100 + si)).save(src, quality=95)
            decoded = ImageLoader.load_as_array(src)
            arrays[f"in_{si}"] = decoded
            for op in ops:
                for seed in seeds:
                    if (h, w) == (224, 224) and seed != 7:
                        continue  # keep the fixture small
                    aug = ImageAugmenter(seed=seed)
                    ok = getattr(aug, op)(str(src), str(td / "out.jpg"))
                    assert ok, (op, seed)
                    key = f"out_{si}_{op}_{seed}"
                    arrays[key] = captured["img"]
                    case = {"input": f"in_{si}", "op": op, "seed": seed, "output": key}
                    # distortion's noise is numpy's own stream (np.random.seed(seed);
                    # np.random.normal(0, 5, shape), image_augmenter.py:121): tests re-draw it.
                    cases.append(case)

        # loader path: resize_image (LANCZOS) + normalize_array (sequence.py:83-89)
        for si, (h, w) in enumerate(sizes):
            img = Image.fromarray(arrays[f"in_{si}"])
            for S in (32, 64, 224):
                r = np.array(ImageTransforms.resize_image(img, (S, S)))
                arrays[f"resize_{si}_{S}"] = r
            if si < 2:
                arrays[f"norm_{si}"] = ImageTransforms.normalize_array(arrays[f"in_{si}"])

    np.savez_compressed(OUT / "augment_golden.npz", **arrays)

    # integer-exact host logic
    planner_counts = {
        "Apple": {"Apple_healthy": 1640, "Apple_scab": 629, "Apple_rust": 275, "Apple_Black_rot": 620},
        "Grape": {"Grape_healthy": 422, "Grape_spot": 1075, "Grape_Esca": 1382, "Grape_Black_rot": 1178},
    }
    plan = AugmentationPlanner(planner_counts).calculate_plan()
    items = [ManifestItem(id=str(i), plant="p", cls="c", label=lab, split="train", src=Path("x"))
             for i, lab in enumerate(["b__z", "a__y", "b__z", "c__x", "a__y"])]
    rng = random.Random(3)
    yt = [rng.randrange(4) for _ in range(200)]
    yp = [rng.randrange(4) for _ in range(200)]
    meta = {
        "cases": cases,
        "planner": {"counts": planner_counts, "plan": plan},
        "label_mapping": {"labels": [it.label for it in items],
                          "label2idx": build_label_mapping(items)},
        "confusion": {"y_true": yt, "y_pred": yp, "num_classes": 4,
                      "matrix": compute_confusion_counts(yt, yp, 4)},
    }
    (OUT / "augment_golden.json").write_text(json.dumps(meta, indent=1))
    print("wrote", OUT / "augment_golden.npz", (OUT / "augment_golden.npz").stat().st_size, "bytes;",
          len(cases), "cases")


if __name__ == "__main__":
    main()
