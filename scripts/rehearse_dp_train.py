"""Rehearsal of `python -m torch.distributed.run ... -m leaffliction_amd.cli.train` with two ranks on
a one-GPU box (LEAFFLICTION_DIST_BACKEND=gloo: the ranks share the card, collectives through gloo).
Builds a two-class toy dataset, trains 2 epochs at global batch 16 with world size 2 and again with
world size 1, and prints both histories: same data order, same global batch, so the loss curves
must agree up to BatchNorm's per-rank statistics (8 vs 16 images per normalisation)."""
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent


def colour_tree(root: Path, n_per_class, size):
    rng = np.random.RandomState(0)
    for cls, col in (("Apple_healthy", (60, 140, 50)), ("Apple_rust", (150, 80, 30))):
        d = root / "Apple" / cls
        d.mkdir(parents=True)
        for i in range(n_per_class):
            img = np.clip(rng.normal(0, 12, (size, size, 3)) + np.array(col), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(d / f"image ({i + 1}).JPG", quality=95)


def manifest(root: Path, out: Path):
    items = []
    for class_dir in sorted((root / "Apple").iterdir()):
        for i, f in enumerate(sorted(class_dir.glob("*.JPG"))):
            items.append({"plant": "Apple", "class": class_dir.name, "label": f"Apple__{class_dir.name}",
                          "split": "val" if i % 4 == 0 else "train", "src": str(f.resolve()),
                          "id": f"Apple/{class_dir.name}/{f.name}"})
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps({"meta": {"seed": 32}, "items": items}))


def run(world: int, work: Path, man: Path):
    env = dict(os.environ, PYTHONPATH=str(ROOT), LEAFFLICTION_DIST_BACKEND="gloo")
    args = ["-m", "leaffliction_amd.cli.train", "--manifest", str(man), "--epochs", "2", "--batch-size", "16",
            "--img-size", "64", "--no-mixed-precision", "--seed", "42"]
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29544"] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(r.stdout[-3000:], r.stderr[-3000:])
        raise SystemExit(f"world {world}: exit code {r.returncode}")
    mdir = work / "artifacts/models"
    hist = json.loads((mdir / "history.json").read_text())
    meta = json.loads((mdir / "meta.json").read_text())
    cm = json.loads((mdir / "confusion_matrix.json").read_text())
    return hist, meta, cm


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        colour_tree(td / "images", 32, 80)
        out = {}
        for world in (2, 1):
            work = td / f"w{world}"
            work.mkdir()
            man = work / "artifacts/datasets/manifest_split.json"
            manifest(td / "images", man)
            hist, meta, cm = run(world, work, man)
            out[world] = hist
            n_val = int(np.array(cm["matrix"]).sum())
            print(f"world {world}: loss {[round(v, 4) for v in hist['loss']]} val_loss "
                  f"{[round(v, 4) for v in hist['val_loss']]} val_acc {hist['val_accuracy']} "
                  f"confusion total {n_val} meta.world_size {meta.get('parallel', meta).get('world_size', '?')}",
                  flush=True)
        d = max(abs(a - b) for a, b in zip(out[1]["loss"], out[2]["loss"]))
        print(f"max |loss(world 1) - loss(world 2)| = {d:.4f}")
