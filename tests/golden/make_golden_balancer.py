"""Golden vectors for the dataset-balancing loop, produced by running the REFERENCE's
`DatasetBalancer` (srcs/preprocessing/dataset_balancer.py) in this container.

The reference fans tasks out over a ProcessPoolExecutor; here the pool is replaced by an
in-process stand-in so that the reference's own task-building code and its own
`_process_single_transformation` run deterministically and every task can be recorded.
Directory listing order is filesystem-dependent, so the image lists
are sorted; the GPU test applies the same sort.  Only data is written: balancer_golden.*.
"""
from __future__ import annotations

import json
import sys
import tempfile
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REF))
sys.path.insert(0, str(OUT))

from make_golden import leaf_like  # noqa: E402
from srcs.preprocessing import dataset_balancer as DB  # noqa: E402

LAYOUT = {"Apple": {"Apple_healthy": 9, "Apple_scab": 2}}
SIZE = 48


def build_dataset(root: Path) -> None:
    k = 0
    for plant, classes in LAYOUT.items():
        for cls, n in classes.items():
            d = root / plant / cls
            d.mkdir(parents=True)
            for i in range(n):
                Image.fromarray(leaf_like(SIZE, SIZE, 500 + k)).save(d / f"image ({i + 1}).JPG", quality=95)
                k += 1


class _Future:
    def __init__(self, value):
        self._v = value

    def result(self):
        return self._v


class _InlinePool:
    recorded = []

    def __init__(self, max_workers=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def submit(self, fn, task):
        _InlinePool.recorded.append(dict(task))
        return _Future(fn(task))


def main() -> None:
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        src, dst = td / "images", td / "augmented"
        build_dataset(src)
        DB.ProcessPoolExecutor = _InlinePool
        DB.as_completed = lambda futs: list(futs)
        bal = DB.DatasetBalancer(source_dir=str(src), target_dir=str(dst), seed=42, workers=1)
        orig = bal._get_images_by_class
        bal._get_images_by_class = lambda: {k: sorted(v) for k, v in orig().items()}
        bal._generate_augmented_manifest = lambda: None
        bal.analyze_distribution()
        plan = bal.calculate_plan()
        bal.execute_balancing()
        tasks, arrays = [], {}
        for i, t in enumerate(_InlinePool.recorded):
            out = Path(t["output_path"])
            arrays[f"out_{i}"] = np.array(Image.open(out).convert("RGB"))
            tasks.append({"source": Path(t["source_img"]).name, "output": out.name,
                          "transform": t["transform_name"], "class": t["class_name"],
                          "seed": t["seed"], "array": f"out_{i}",
                          "sha_bytes": __import__("hashlib").sha1(out.read_bytes()).hexdigest()})
    np.savez_compressed(OUT / "balancer_golden.npz", **arrays)
    (OUT / "balancer_golden.json").write_text(json.dumps(
        {"layout": LAYOUT, "size": SIZE, "seed": 42, "plan": plan, "tasks": tasks}, indent=1))
    print("wrote", len(tasks), "tasks")


if __name__ == "__main__":
    main()
