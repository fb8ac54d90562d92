"""Convergence evidence for BASELINE north_star's "val-accuracy within +-0.5 %" bar, read as a bf16-vs-fp32 proxy.

The Keras delta itself cannot be measured here (keras / tensorflow are not installable in this image, and the
reference ships no trained checkpoint or accuracy figure); what can be measured is whether the mixed-precision
step (bf16 storage and MFMA operands) trains the same model to the same validation accuracy as the fp32 step,
through the product's own `cli.train` entry point (srcs/cli/train.py:389-447 -> leaffliction_amd/cli/train.py),
on the same files, seed, schedule and epochs.

  python scripts/convergence.py [--images 10000] [--epochs 10] [--batch 256] [--out profiles/r03_convergence.json]

Dataset: synthetic labelled leaves, 8 classes = 4 disc colours x {few, many} lesions, 224 x 224 JPEGs (quality 95)
with per-image jitter of position, radius, colour and background noise — learnable, not trivial (the colour
classes overlap under the jitter, the lesion classes need spatial evidence).  80 % train / 20 % val, stratified.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
COLOURS = [(60, 140, 50), (95, 130, 45), (150, 140, 50), (120, 95, 40)]   # green, olive, yellowing, browning
CLASSES = [f"c{c}{'m' if m else 'f'}" for c in range(4) for m in (0, 1)]


def make_image(job):
    path, cls, seed, size = job
    from PIL import Image
    rng = np.random.RandomState(seed)
    colour, many = divmod(cls, 2)
    h = w = size
    img = np.clip(rng.normal(150, 10, (h, w, 1)).repeat(3, axis=2) + rng.normal(0, 6, (1, 1, 3)), 0, 255)
    yy, xx = np.mgrid[0:h, 0:w]
    s = size / 224.0
    cy, cx = rng.randint(int(85 * s), int(140 * s) + 1, 2)
    ry, rx = rng.randint(int(50 * s), int(85 * s) + 1, 2)
    leaf = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    base = np.array(COLOURS[colour], np.float64) + rng.normal(0, 9, 3)          # the colour classes overlap
    img[leaf] = base
    n_spots = rng.randint(6, 14) if many else rng.randint(0, 4)
    for _ in range(n_spots):
        by, bx = cy + rng.randint(-ry, ry + 1), cx + rng.randint(-rx, rx + 1)
        br = rng.randint(max(2, int(3 * s)), max(3, int(8 * s)) + 1)
        spot = ((yy - by) ** 2 + (xx - bx) ** 2 <= br * br) & leaf
        img[spot] = np.array((105, 65, 30), np.float64) + rng.normal(0, 8, 3)
    img = img + rng.normal(0, 8, img.shape)
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(path, quality=95)
    return path


def build_dataset(root: Path, n: int, size: int) -> Path:
    from concurrent.futures import ProcessPoolExecutor
    jobs, items = [], []
    per = n // len(CLASSES)
    for ci, cname in enumerate(CLASSES):
        d = root / "images" / "Leaf" / cname
        d.mkdir(parents=True)
        for i in range(per):
            p = d / f"img_{i:05d}.JPG"
            jobs.append((str(p), ci, 1000003 * ci + i, size))
            items.append({"id": f"Leaf/{cname}/{p.name}", "plant": "Leaf", "class": cname, "label": f"Leaf__{cname}",
                          "split": "val" if i % 5 == 4 else "train", "src": str(p)})
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        list(ex.map(make_image, jobs, chunksize=64))
    man = root / "artifacts" / "datasets" / "manifest_split.json"
    man.parent.mkdir(parents=True)
    man.write_text(json.dumps({"meta": {"generator": "scripts/convergence.py", "images": len(items)}, "items": items}))
    return man


def train(root: Path, manifest: Path, tag: str, epochs: int, batch: int, size: int, fp32: bool) -> dict:
    work = root / tag
    work.mkdir()
    cmd = [sys.executable, "-m", "leaffliction_amd.cli.train", "--manifest", str(manifest), "--epochs", str(epochs),
           "--batch-size", str(batch), "--img-size", str(size), "--seed", "42", "--base"]
    if fp32:
        cmd.append("--no-mixed-precision")
    env = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.time()
    log = (work / "train.log").open("w")
    rc = subprocess.run(cmd, cwd=work, env=env, stdout=log, stderr=subprocess.STDOUT).returncode
    sec = time.time() - t0
    out = work / "artifacts" / "models"
    hist = json.loads((out / "history.json").read_text())
    meta = json.loads((out / "meta.json").read_text())
    cm = json.loads((out / "confusion_matrix.json").read_text())
    counts = np.array(cm.get("matrix", cm.get("counts", cm if isinstance(cm, list) else [])))
    final_acc = float(np.trace(counts) / counts.sum()) if counts.size else None
    return {"rc": rc, "seconds": round(sec, 1), "mixed_precision": meta["training"]["mixed_precision"],
            "saved_variant": meta["saved_variant"], "history": {k: [round(v, 5) for v in vs] for k, vs in hist.items()},
            "val_accuracy_last_epoch": hist["val_accuracy"][-1], "val_accuracy_best_epoch": max(hist["val_accuracy"]),
            "saved_model_val_accuracy": final_acc}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=10000)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--out", type=Path, default=ROOT / "profiles" / "r03_convergence.json")
    args = ap.parse_args()
    root = Path(tempfile.mkdtemp(prefix="lf_conv_"))
    try:
        t0 = time.time()
        manifest = build_dataset(root, args.images, args.size)
        print(f"dataset: {args.images} images in {time.time() - t0:.1f} s", flush=True)
        runs = {}
        for tag, fp32 in (("fp32", True), ("bf16", False)):
            runs[tag] = train(root, manifest, tag, args.epochs, args.batch, args.size, fp32)
            print(tag, "val_accuracy per epoch:", runs[tag]["history"]["val_accuracy"], f"({runs[tag]['seconds']} s)",
                  flush=True)
        n_val = args.images // len(CLASSES) // 5 * len(CLASSES)
        doc = {"what": "cli.train, base preset (widths 32/64/128/256, AdamW + cosine decay + clipnorm + EMA, label "
                       "smoothing 0.02, in-model augmentation, dropout), same files / seed / epochs, fp32 step vs "
                       "mixed-precision (bf16) step; the Keras reference's own accuracy cannot be measured in this image "
                       "(keras / tensorflow absent), so the +-0.5 % bar is read as bf16 vs fp32",
               "dataset": {"images": args.images, "train": args.images - n_val, "val": n_val, "classes": len(CLASSES),
                           "img_size": args.size, "generator": "scripts/convergence.py (synthetic labelled leaves)"},
               "epochs": args.epochs, "batch_size": args.batch, "runs": runs,
               "delta_val_accuracy_saved_model": round(runs["bf16"]["saved_model_val_accuracy"] -
                                                       runs["fp32"]["saved_model_val_accuracy"], 5),
               "delta_val_accuracy_last_epoch": round(runs["bf16"]["val_accuracy_last_epoch"] -
                                                      runs["fp32"]["val_accuracy_last_epoch"], 5),
               "one_val_image_is": round(1.0 / n_val, 5)}
        args.out.parent.mkdir(parents=True, exist_ok=True)
        args.out.write_text(json.dumps(doc, indent=1))
        print(json.dumps({k: v for k, v in doc.items() if k.startswith("delta") or k == "one_val_image_is"}))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
