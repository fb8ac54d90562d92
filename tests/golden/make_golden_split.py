"""Golden vectors for the train/val split, produced by running the REFERENCE's split module
(srcs/cli/split.py) in this container on a synthetic directory tree.  The split never opens
an image, so the files are empty; only names and counts matter.  Only data is written:
split_golden.json (layout, scan order, both allocation strategies, the seeded split map, the
summary rows, and the Distribution CLI's counts / merged CSVs on the same tree)."""
from __future__ import annotations

import csv
import json
import sys
import tempfile
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REF))

from srcs.cli import Distribution as D  # noqa: E402
from srcs.cli import split as S  # noqa: E402

LAYOUT = {"Apple": {"healthy": 7, "rust": 5, "scab": 1}, "Grape": {"esca": 12, "healthy": 3, "empty": 0}}


def build(root: Path) -> None:
    for plant, classes in LAYOUT.items():
        for cls, n in classes.items():
            d = root / plant / cls
            d.mkdir(parents=True)
            for i in range(n):
                (d / f"image ({i + 1}).{'JPG' if i % 2 else 'jpg'}").write_bytes(b"")
            (d / "notes.txt").write_text("not an image")
            (d / "picture.png").write_bytes(b"")


def main() -> None:
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp) / "images"
        build(root)
        items = S.scan_dataset(root)
        by_label = {}
        for it in items:
            by_label.setdefault(it.label, []).append(it)
        counts = {k: len(v) for k, v in by_label.items()}
        out = {"layout": LAYOUT, "scan": [it.rel_id for it in items], "counts": counts, "cases": []}
        for name, alloc in (("ratio_0.2", S.allocate_validation_by_ratio(counts, 0.2)),
                            ("ratio_0.5", S.allocate_validation_by_ratio(counts, 0.5)),
                            ("min_val_6", S.allocate_validation_counts(counts, 6)),
                            ("min_val_100", S.allocate_validation_counts(counts, 100)),
                            ("min_val_0", S.allocate_validation_counts(counts, 0))):
            for seed in (32, 7):
                sm = S.build_split_map(by_label, alloc, seed)
                summ = Path(tmp) / "s.csv"
                S.write_summary(summ, by_label, sm)
                with summ.open() as f:
                    rows = list(csv.reader(f))
                out["cases"].append({"name": name, "seed": seed, "alloc": alloc, "split": sm, "summary": rows})
        man = Path(tmp) / "m.json"
        S.write_manifest(man, items, sm, src_root=root, seed=7, min_val=0)
        doc = json.loads(man.read_text())
        out["manifest_meta_keys"] = sorted(doc["meta"])
        out["manifest_item_keys"] = list(doc["items"][0])
        out["manifest_strategy"] = doc["meta"]["strategy"]
        # Distribution.py: counts (all plants / one plant), CSV merge over an existing file,
        # and replacement of a file with a foreign header
        rows_all = D.count_images(root, None)
        rows_g = D.count_images(root, {"Grape"})
        p = Path(tmp) / "d.csv"
        p.write_text("plant,class,count\nApple,healthy,99\nPear,ripe,4\n")
        D.merge_csv(rows_g, p)
        with p.open() as f:
            merged = list(csv.reader(f))
        p2 = Path(tmp) / "e.csv"
        p2.write_text("a,b\n1,2\n")
        D.merge_csv(rows_all, p2)
        with p2.open() as f:
            fresh = list(csv.reader(f))
        out["distribution"] = {"rows_all": rows_all, "rows_grape": rows_g, "merged": merged, "fresh": fresh}
    (OUT / "split_golden.json").write_text(json.dumps(out, indent=1, sort_keys=True))
    print("wrote", OUT / "split_golden.json", len(out["scan"]), "items")


if __name__ == "__main__":
    main()
