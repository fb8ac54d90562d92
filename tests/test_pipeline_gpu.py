"""End-to-end entrypoints on the GPU: Augmentation (balancer golden), train (config C1:
2 classes, img 64, batch 8, 1 epoch), predict (batch + JSON schema + accuracy gate)."""
import json
import os
import random
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

from conftest import leaf_like

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def build_tree(root: Path, layout, size, seed0):
    k = 0
    for plant, classes in layout.items():
        for cls, n in classes.items():
            d = root / plant / cls
            d.mkdir(parents=True)
            for i in range(n):
                Image.fromarray(leaf_like(size, size, seed0 + k)).save(d / f"image ({i + 1}).JPG", quality=95)
                k += 1


def test_balancer_outputs_match_reference_pixels(cuda, tmp_path, monkeypatch):
    """GPU DatasetBalancer == the reference's DatasetBalancer: names, seeds and decoded pixels."""
    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    gold = json.loads((GOLD / "balancer_golden.json").read_text())
    arrays = np.load(GOLD / "balancer_golden.npz")
    monkeypatch.chdir(tmp_path)
    src, dst = tmp_path / "images", tmp_path / "augmented"
    build_tree(src, gold["layout"], gold["size"], 500)
    bal = DatasetBalancer(source_dir=str(src), target_dir=str(dst), seed=gold["seed"], workers=2)
    orig = bal._images_by_class
    bal._images_by_class = lambda: {k: sorted(v) for k, v in orig().items()}
    bal.run()
    assert bal.completed == len(gold["tasks"]) and bal.failed == 0
    for t in gold["tasks"]:
        out = dst / "Apple" / t["class"] / t["output"]
        assert out.exists(), t
        got = np.array(Image.open(out).convert("RGB"))
        assert np.array_equal(got, arrays[t["array"]]), t
    man = json.loads((tmp_path / "artifacts/datasets/manifest_augmented.json").read_text())
    assert man["meta"]["augmented_images"] == len(gold["tasks"]) and man["meta"]["augmentation_seed"] == 42
    assert all(it["split"] == "train" and it["label"] == f"{it['plant']}__{it['class']}" for it in man["items"])


def test_augmentation_cli_single_image(cuda, tmp_path, monkeypatch):
    from leaffliction_amd.cli import Augmentation
    monkeypatch.chdir(tmp_path)
    img = tmp_path / "leaf.jpg"
    Image.fromarray(leaf_like(64, 64, 1)).save(img, quality=95)
    Augmentation.main([str(img), "-out", str(tmp_path / "ex"), "-seed", "42"])
    names = sorted(p.name for p in (tmp_path / "ex").iterdir())
    assert names == sorted(["original_leaf.jpg"] + [f"{t}_leaf.jpg" for t in Augmentation.TRANSFORMATIONS])
    with pytest.raises(SystemExit) as e:
        Augmentation.main([str(tmp_path / "nope")])
    assert e.value.code == 1


def write_split_manifest(root: Path, out: Path, val_every=4):
    items = []
    for plant_dir in sorted(root.iterdir()):
        for class_dir in sorted(plant_dir.iterdir()):
            for i, f in enumerate(sorted(class_dir.glob("*.JPG"))):
                items.append({"plant": plant_dir.name, "class": class_dir.name,
                              "label": f"{plant_dir.name}__{class_dir.name}",
                              "split": "val" if i % val_every == 0 else "train",
                              "src": str(f.resolve()), "id": f"{plant_dir.name}/{class_dir.name}/{f.name}"})
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps({"meta": {"seed": 32}, "items": items}))


def colour_tree(root: Path, n_per_class, size):
    """Two trivially separable classes (green vs brown leaves) so 1 epoch learns something."""
    rng = np.random.RandomState(0)
    for cls, col in (("Apple_healthy", (60, 140, 50)), ("Apple_rust", (150, 80, 30))):
        d = root / "Apple" / cls
        d.mkdir(parents=True)
        for i in range(n_per_class):
            img = np.clip(rng.normal(0, 12, (size, size, 3)) + np.array(col), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(d / f"image ({i + 1}).JPG", quality=95)


def test_train_and_predict_cli_config_c1(cuda, tmp_path, monkeypatch):
    """BASELINE configs[0]: Apple subset, 2 classes, img-size 64, batch 8, 1 epoch — plumbing."""
    from leaffliction_amd.cli import predict as predict_cli
    from leaffliction_amd.cli import train as train_cli
    monkeypatch.chdir(tmp_path)
    colour_tree(tmp_path / "images", 24, 80)   # 80x80 sources: exercises the GPU LANCZOS resize
    man = tmp_path / "artifacts/datasets/manifest_split.json"
    write_split_manifest(tmp_path / "images", man)
    train_cli.main(["--manifest", str(man), "--epochs", "2", "--batch-size", "8", "--img-size", "64",
                    "--no-mixed-precision", "--seed", "42"])
    mdir = tmp_path / "artifacts/models"
    for f in ("leaf_cnn.keras", "labels.json", "history.json", "meta.json", "confusion_matrix.json",
              "confusion_matrix.png"):
        assert (mdir / f).exists(), f
    labels = json.loads((mdir / "labels.json").read_text())["label2idx"]
    assert labels == {"Apple__Apple_healthy": 0, "Apple__Apple_rust": 1}
    hist = json.loads((mdir / "history.json").read_text())
    assert set(hist) == {"loss", "accuracy", "val_loss", "val_accuracy", "learning_rate"}
    assert len(hist["loss"]) == 2 and all(np.isfinite(hist["loss"]))
    meta = json.loads((mdir / "meta.json").read_text())
    assert meta["data"]["img_size"] == 64 and meta["model"]["widths"] == [32, 64, 128, 256]
    assert meta["saved_variant"] in ("base", "ema") and meta["labels"] == sorted(labels, key=labels.get)
    cm = json.loads((mdir / "confusion_matrix.json").read_text())
    n_val = sum(1 for it in json.loads(man.read_text())["items"] if it["split"] == "val")
    assert np.array(cm["matrix"]).sum() == n_val and cm["labels"] == meta["labels"]
    # missing manifest: logs and returns (exit code 0), like the reference
    train_cli.main(["--manifest", str(tmp_path / "nope.json"), "--epochs", "1"])
    # --fast preset (Adam, sparse labels, cached loader) and the small / tiny scales run too
    for extra in (["--fast", "--small"], ["--tiny", "--no-normalization"]):
        train_cli.main(["--manifest", str(man), "--epochs", "1", "--batch-size", "8", "--img-size", "64",
                        "--no-mixed-precision", "--seed", "3"] + extra)
        h2 = json.loads((mdir / "history.json").read_text())
        assert len(h2["loss"]) == 1 and np.isfinite(h2["loss"][0]) and np.isfinite(h2["val_loss"][0])
    train_cli.main(["--manifest", str(man), "--epochs", "2", "--batch-size", "8", "--img-size", "64",
                    "--no-mixed-precision", "--seed", "42"])   # restore the base model for predict

    # predict: batch mode -> JSON schema
    predict_cli.main([str(tmp_path / "images/Apple/Apple_rust"), "-batch", "-learnings", str(mdir),
                      "-json", "artifacts/prediction_output/batch_results.json"])
    out = json.loads((tmp_path / "artifacts/prediction_output/batch_results.json").read_text())
    # 24 files, listed twice: the reference's get_image_files globs "*.JPG" and "**/*.JPG"
    # (image_utils.py:81-88) — the quirk is part of the contract
    assert set(out) == {"batch_results", "summary"} and out["summary"]["total_images"] == 48
    r0 = out["batch_results"][0]
    assert set(r0) == {"image_path", "top_prediction", "confidence", "all_probabilities"}
    assert abs(sum(r0["all_probabilities"].values()) - 1.0) < 1e-4
    # evaluate gate: unreachable target -> exit code 2, nothing emitted
    with pytest.raises(SystemExit) as e:
        predict_cli.main([str(tmp_path / "images"), "-batch", "--evaluate", "--manifest", str(man),
                          "--split", "val", "--sample-size", "6", "--target-acc", "1.01",
                          "--max-attempts", "2", "-learnings", str(mdir)])
    assert e.value.code == 2
    with pytest.raises(SystemExit) as e:
        predict_cli.main([str(tmp_path / "missing.jpg"), "-learnings", str(mdir)])
    assert e.value.code == 1


def test_fit_learns_separable_classes(cuda, tmp_path):
    """Accuracy on two trivially separable classes reaches 100% within a few epochs (fit loop,
    callbacks, EMA variant selection)."""
    from leaffliction_amd.dataio.manifest import load_manifest, select_items, build_label_mapping
    from leaffliction_amd.dataio.sequence import ManifestSequence
    from leaffliction_amd.model.cnn import build_leafcnn, adapt_normalization
    from leaffliction_amd.train.utils import (CosineDecay, build_callbacks, build_loss, build_optimizer,
                                              save_best_variant)
    colour_tree(tmp_path / "images", 20, 32)
    man = tmp_path / "m.json"
    write_split_manifest(tmp_path / "images", man)
    items = load_manifest(man)
    tr, va = select_items(items, "train"), select_items(items, "val")
    l2i = build_label_mapping(tr)
    cfg = {"optimizer": "adamw", "lr": 2e-3, "weight_decay": 1e-4, "label_smoothing": 0.02,
           "cosine_decay": True, "ema_decay": 0.999, "clipnorm": 0.5}
    tseq = ManifestSequence(tr, l2i, 32, 8, True, 42, num_classes=2, one_hot=True)
    vseq = ManifestSequence(va, l2i, 32, 8, False, 42, num_classes=2, one_hot=True, cache=True)
    model, norm = build_leafcnn(num_classes=2, img_size=32, widths=[16, 32, 64], drop_block=0.1,
                                drop_top=0.3, l2_reg=1e-4, seed=1)
    adapt_normalization(norm, tseq)
    assert norm.adapted and norm.mean.shape == (3,)
    model.compile(build_optimizer(cfg, CosineDecay(2e-3, len(tseq) * 6)), build_loss(cfg), ["accuracy"])
    cbs, ema = build_callbacks(cfg)
    hist = model.fit(tseq, validation_data=vseq, epochs=6, callbacks=cbs, verbose=0)
    assert hist.history["accuracy"][-1] > 0.9
    variant = save_best_variant(model, vseq, ema, tmp_path / "out", l2i, hist, meta={"x": 1})
    assert variant in ("base", "ema")
    cm = json.loads((tmp_path / "out/confusion_matrix.json").read_text())["matrix"]
    assert cm[0][0] + cm[1][1] == len(va)      # after best-variant selection: all correct


def test_device_resident_cache_batches_equal_host_loader(cuda, tmp_path):
    """cache=True keeps the resized uint8 dataset in HBM and assembles batches with the gather
    kernel: same pixels, same labels, same shuffling and rank slicing as the uncached loader."""
    import torch
    from leaffliction_amd.dataio.manifest import load_manifest, select_items, build_label_mapping
    from leaffliction_amd.dataio.sequence import ManifestSequence
    colour_tree(tmp_path / "images", 14, 40)   # native 40x40 -> resized to 32 on the GPU
    man = tmp_path / "m.json"
    write_split_manifest(tmp_path / "images", man)
    items = load_manifest(man)
    tr = select_items(items, "train")
    l2i = build_label_mapping(tr)
    for world in (1, 2):
        for rank in range(world):
            a = ManifestSequence(tr, l2i, 32, 6, True, 7, num_classes=2, one_hot=True, rank=rank, world=world)
            b = ManifestSequence(tr, l2i, 32, 6, True, 7, num_classes=2, one_hot=True, cache=True,
                                 rank=rank, world=world)
            assert b._cache_dev is not None and tuple(b._cache_dev.shape) == (len(tr), 32, 32, 3)
            for epoch in range(2):
                for i in range(len(a)):
                    (xa, ya), (xb, yb) = a[i], b[i]
                    assert xb.is_cuda and torch.equal(xa, xb) and np.array_equal(ya, yb)
                a.on_epoch_end()
                b.on_epoch_end()


def test_five_step_workflow_distribution_augment_split_train_predict(cuda, tmp_path, monkeypatch):
    """The reference README's workflow end to end with this repo's entrypoints only:
    Distribution -> Augmentation (balance) -> split -> train -> predict --evaluate."""
    import csv
    from leaffliction_amd.cli import Augmentation, Distribution, predict as predict_cli, split, train as train_cli
    monkeypatch.chdir(tmp_path)
    root = tmp_path / "images" / "Apple"
    rng = np.random.RandomState(3)
    for cls, colour, n in (("Apple_healthy", (40, 170, 60), 14), ("Apple_rust", (170, 80, 40), 6)):
        d = root / cls
        d.mkdir(parents=True)
        for i in range(n):
            img = np.clip(np.array(colour)[None, None] + rng.normal(0, 12, (72, 72, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(d / f"image ({i + 1}).JPG", quality=95)
    Distribution.main([str(tmp_path / "images"), "--no-plots"])
    with (tmp_path / "artifacts/plots/distribution.csv").open() as f:
        assert list(csv.reader(f))[1:] == [["Apple", "Apple_healthy", "14"], ["Apple", "Apple_rust", "6"]]
    Augmentation.main([str(tmp_path / "images"), "-seed", "42"])
    aug = tmp_path / "artifacts/augmented_directory"
    counts = {d.name: len(list(d.glob("*.JPG"))) for d in (aug / "Apple").iterdir()}
    assert counts == {"Apple_healthy": 14, "Apple_rust": 14}          # balanced up to the largest class
    man = tmp_path / "artifacts/datasets/manifest_split.json"
    split.main(["--src", str(aug), "--out", str(man.parent), "--out-manifest", str(man), "--val-ratio", "0.25"])
    items = json.loads(man.read_text())["items"]
    assert len(items) == 28 and sum(it["split"] == "val" for it in items) == 8
    train_cli.main(["--manifest", str(man), "--epochs", "4", "--batch-size", "8", "--img-size", "32",
                    "--no-mixed-precision", "--seed", "1", "--tiny"])
    mdir = tmp_path / "artifacts/models"
    hist = json.loads((mdir / "history.json").read_text())
    assert hist["val_accuracy"][-1] >= 0.75
    predict_cli.main([str(aug), "-batch", "--evaluate", "--manifest", str(man), "--split", "val",
                      "--sample-size", "8", "--target-acc", "0.7", "--max-attempts", "3", "-learnings", str(mdir)])
    ev = json.loads((tmp_path / "artifacts/prediction_output/evaluation/evaluation_results.json").read_text())
    assert ev["metrics"]["accuracy"] >= 0.7 and ev["evaluation_info"]["valid_predictions"] == 8


def test_transform_filters_host_mirror(cuda):
    """apply_blur_filter / analyze_color_regions with the reference's call shapes (blur.py:18-20,
    hist.py:22-24,188-189) against the oracle."""
    import numpy as np
    from conftest import leaf_like
    from leaffliction_amd.transform import TransformConfig, analyze_color_regions, apply_blur_filter, hue_range_counts
    from leaffliction_amd.transform.filters import REGION_KEYS, HUE_KEYS
    from oracle import cv_ops as CV
    img = leaf_like(96, 80, 4)
    yy, xx = np.mgrid[0:96, 0:80]
    mask = (((yy - 48) ** 2 + (xx - 40) ** 2) <= 30 ** 2).astype(np.uint8) * 255
    cfg = TransformConfig()
    out = apply_blur_filter(img, cfg, lambda rgb: (mask, None))
    assert np.array_equal(out, CV.blur_saliency(img, mask))
    out3 = apply_blur_filter(img, cfg, lambda rgb: (np.repeat(mask[..., None], 3, 2), None))   # 3-channel mask
    assert np.array_equal(out3, out)
    assert apply_blur_filter(img, cfg, lambda rgb: (None, None)) is img                        # blur.py:22-24

    class NoBrown:
        gaussian_sigma = 1.5
    assert np.array_equal(apply_blur_filter(img, NoBrown(), lambda rgb: (mask, None)),
                          CV.blur_saliency(img, mask, use_brown=False))
    counts, _ = CV.hsv_region_stats(img)
    got = analyze_color_regions(img)
    assert list(got) == list(REGION_KEYS)
    for i, k in enumerate(REGION_KEYS):
        assert got[k] == (int(counts[1 + i]) / int(counts[0])) * 100
    assert hue_range_counts(img) == {k: int(counts[9 + i]) for i, k in enumerate(HUE_KEYS)}
    assert analyze_color_regions(np.zeros((8, 8, 3), np.uint8)) == {}
    # the density curves matplotlib would draw (hist.py:153-167): numpy's histogram of the leaf
    # pixels of each channel over its own range
    from leaffliction_amd.transform import hsv_density_curves
    hsv = CV.rgb2hsv(img).astype(np.int64)
    leaf = (hsv[..., 1] > 10) & (hsv[..., 2] > 15) & (hsv[..., 2] < 245)
    curves = hsv_density_curves(img)
    for ci, name in enumerate("HSV"):
        want_d, want_e = np.histogram(hsv[..., ci][leaf], bins=60, density=True)
        got_d, got_e = curves[name]
        assert np.array_equal(got_e, want_e) and np.allclose(got_d, want_d, rtol=1e-12, atol=0)


def test_predict_in_bf16_mode_gives_the_same_confusion_matrix(cuda, tmp_path, monkeypatch):
    """BASELINE configs[4]: reduced-precision inference reports the same confusion matrix as fp32
    (train on the two-colour toy set, then predict every validation image both ways)."""
    from leaffliction_amd.cli import train as train_cli
    from leaffliction_amd.predict.predictor import Predictor
    from leaffliction_amd.utils.confusion_matrix import compute_confusion_counts
    monkeypatch.chdir(tmp_path)
    colour_tree(tmp_path / "images", 24, 64)
    man = tmp_path / "artifacts/datasets/manifest_split.json"
    write_split_manifest(tmp_path / "images", man)
    train_cli.main(["--manifest", str(man), "--epochs", "2", "--batch-size", "8", "--img-size", "64",
                    "--no-mixed-precision", "--seed", "42"])
    items = [it for it in json.loads(man.read_text())["items"] if it["split"] == "val"]
    paths = [it["src"] for it in items]
    results = {}
    for mode in ("f32", "bf16"):
        monkeypatch.setenv("LEAFFLICTION_INFER_DTYPE", mode)
        pred = Predictor(tmp_path / "artifacts/models")
        pred.load()
        assert pred.model_loader.model.infer_dtype == mode
        out = pred.predict_batch(paths)
        results[mode] = out
    labels = sorted({it["label"] for it in items})
    idx = {lab: i for i, lab in enumerate(labels)}
    y_true = [idx[it["label"]] for it in items]
    cms = {}
    for mode, out in results.items():
        y_pred = [idx[r["top_prediction"]] for r in out]
        cms[mode] = compute_confusion_counts(y_true, y_pred, len(labels))
    assert np.array_equal(np.array(cms["f32"]), np.array(cms["bf16"]))
    assert np.trace(np.array(cms["bf16"])) == len(items)       # the toy set is separable
    p32 = np.array([list(r["all_probabilities"].values()) for r in results["f32"]])
    p16 = np.array([list(r["all_probabilities"].values()) for r in results["bf16"]])
    assert 0 < np.abs(p32 - p16).max() < 3e-2


def test_pooled_predict_batch_equals_the_sequential_loop(cuda, tmp_path, monkeypatch):
    """predict_batch over >= 64 files decodes in codec worker processes + on the GPU: same results, order and
    skipping as the one-by-one loop — on files it Huffman-decodes itself (whole MCUs), files it leaves to Pillow
    (ragged size, 4:4:4), native sizes that need the LANCZOS resize, and one file that cannot be read."""
    from PIL import Image

    from leaffliction_amd.cli import train as train_cli
    from leaffliction_amd.predict.predictor import Predictor
    monkeypatch.chdir(tmp_path)
    colour_tree(tmp_path / "images", 24, 64)
    man = tmp_path / "artifacts/datasets/manifest_split.json"
    write_split_manifest(tmp_path / "images", man)
    train_cli.main(["--manifest", str(man), "--epochs", "1", "--batch-size", "8", "--img-size", "64",
                    "--no-mixed-precision", "--seed", "42"])
    rng = np.random.RandomState(0)
    files = tmp_path / "batch"
    files.mkdir()
    paths = []
    for i in range(90):
        h, w = [(64, 64), (96, 80), (70, 50), (128, 128)][i % 4]
        a = np.clip(rng.normal(128, 40, (h, w, 3)) + (60 if i % 2 else -60) * np.array([1, -1, 0]), 0, 255).astype(np.uint8)
        p = files / f"im_{i:03d}.jpg"
        Image.fromarray(a).save(p, quality=90, **({"subsampling": 0} if i % 7 == 3 else {}))
        paths.append(str(p))
    (files / "broken.jpg").write_bytes(b"\xff\xd8 not really a jpeg")
    paths.insert(17, str(files / "broken.jpg"))
    # a file damaged inside its scan with the EOI in place (the GPU's Huffman decoder hands it back; Pillow conceals the
    # damage and both loops go on with Pillow's pixels) and one cut short (an error in both loops)
    whole = Path(paths[4]).read_bytes()
    (files / "damaged.jpg").write_bytes(whole[:len(whole) // 2] + whole[-2:])
    (files / "cut.jpg").write_bytes(whole[:-300])
    paths.insert(40, str(files / "damaged.jpg"))
    paths.insert(70, str(files / "cut.jpg"))
    pred = Predictor(tmp_path / "artifacts/models")
    pred.load()
    pooled = pred.predict_batch(paths)
    monkeypatch.setattr(Predictor, "POOL_MIN", 10 ** 9)
    serial = pred.predict_batch(paths)
    assert len(pooled) == len(serial) == 91   # 90 files + the damaged one (Pillow reads it); broken and cut are skipped
    for a, b in zip(pooled, serial):
        assert str(a["image_path"]) == str(b["image_path"]) and a["top_prediction"] == b["top_prediction"]
        assert np.array_equal(a["original_array"], b["original_array"])
        pa, pb = np.array(list(a["all_probabilities"].values())), np.array(list(b["all_probabilities"].values()))
        assert np.abs(pa - pb).max() < 1e-6


def test_pooled_loader_batches_and_cache_equal_the_host_loader(cuda, tmp_path, monkeypatch):
    """Device batches of >= POOL_MIN files (and the HBM cache's filling) are decoded by the codec workers + the
    GPU JPEG back end: same uint8 batches as Pillow decode + host stack, over mixed native sizes, 4:4:4 and
    progressive files; an unreadable file raises as the reference's loader does."""
    import torch
    from PIL import Image

    from leaffliction_amd.dataio.manifest import ManifestItem
    from leaffliction_amd.dataio.sequence import ManifestSequence
    rng = np.random.RandomState(3)
    items = []
    for i in range(150):
        h, w = [(64, 64), (96, 80), (70, 50), (48, 48)][i % 4]
        a = leaf_like(h, w, i) if i % 3 else rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        p = tmp_path / (f"im_{i:03d}.jpg" if i % 11 == 5 else f"im_{i:03d}.JPG")
        Image.fromarray(a).save(p, quality=95, progressive=(i % 11 == 5), **({"subsampling": 0} if i % 7 == 3 else {}))
        items.append(ManifestItem(f"id{i}", "plant", f"c{i % 3}", f"c{i % 3}", "train", p))
    l2i = {"c0": 0, "c1": 1, "c2": 2}
    pooled = ManifestSequence(items, l2i, 48, 75, True, 5, num_classes=3)
    monkeypatch.setattr(ManifestSequence, "POOL_MIN", 10 ** 9)
    host = ManifestSequence(items, l2i, 48, 75, True, 5, num_classes=3)
    monkeypatch.undo()
    assert pooled.POOL_MIN == 64
    for i in range(len(host)):
        (xa, ya), (xb, yb) = host[i], pooled[i]
        assert xb.is_cuda and xb.dtype == torch.uint8 and torch.equal(xa, xb) and np.array_equal(ya, yb)
    assert pooled._decoder is not None
    pooled.close()
    cached = ManifestSequence(items, l2i, 48, 75, False, 5, num_classes=3, cache=True)
    assert cached._decoder is None and torch.equal(
        cached._cache_dev, torch.cat([host._load_dev(list(range(b, min(b + 50, 150)))) for b in range(0, 150, 50)]))
    Path(items[20].src).write_bytes(b"\xff\xd8 not a jpeg")
    with pytest.raises(OSError):
        ManifestSequence(items, l2i, 48, 150, False, 5, num_classes=3)[0]


def test_prefetched_loader_batches_equal_the_host_loader(cuda, tmp_path, monkeypatch):
    """`prefetch(i)` starts batch i on the codec workers and `seq[i]` finishes it on the GPU: same batches as the
    host loader in any access order, also when a prefetched batch is never asked for, when a synchronous batch
    comes in between (which drops the pending ones) and across `on_epoch_end` reshuffles; `fit` trains through it."""
    import torch
    from PIL import Image

    from leaffliction_amd.dataio.manifest import ManifestItem
    from leaffliction_amd.dataio.sequence import ManifestSequence
    rng = np.random.RandomState(5)
    items = []
    for i in range(200):
        h, w = [(64, 64), (80, 96), (48, 48)][i % 3]
        p = tmp_path / f"im_{i:03d}.JPG"
        Image.fromarray(leaf_like(h, w, i)).save(p, quality=95, **({"subsampling": 0} if i % 9 == 4 else {}))
        items.append(ManifestItem(f"id{i}", "plant", f"c{i % 2}", f"c{i % 2}", "train", p))
    l2i = {"c0": 0, "c1": 1}
    ahead = ManifestSequence(items, l2i, 48, 32, True, 11, num_classes=2, one_hot=True)
    host = ManifestSequence(items, l2i, 48, 32, True, 11, num_classes=2, one_hot=True)
    monkeypatch.setattr(host, "prefetch", lambda idx: None)
    for epoch in range(2):
        order = list(range(len(host)))
        random.Random(epoch).shuffle(order)
        for j, bi in enumerate(order):
            if j + 1 < len(order):
                ahead.prefetch(order[j + 1])
            if j == 2:
                ahead.prefetch(order[-1])   # a third one: refused (two pending), asked for synchronously later
            if j == 4:
                ahead._load_dev(list(range(70)))   # a synchronous pooled load in between drops the pending batch
            (xa, ya), (xb, yb) = host[bi], ahead[bi]
            assert xb.is_cuda and torch.equal(xa, xb) and np.array_equal(ya, yb)
        host.on_epoch_end()
        ahead.on_epoch_end()
    assert ahead._decoder is not None and not ahead._ahead
    ahead.close()

    from leaffliction_amd.model.cnn import build_leafcnn
    from leaffliction_amd.train.utils import build_loss, build_optimizer
    calls = []
    seq = ManifestSequence(items, l2i, 48, 32, True, 11, num_classes=2, one_hot=True)
    orig = seq.prefetch
    monkeypatch.setattr(seq, "prefetch", lambda idx: (calls.append(idx), orig(idx))[1])
    cfg = {"optimizer": "adamw", "lr": 1e-3, "weight_decay": 1e-4, "label_smoothing": 0.02,
           "cosine_decay": False, "ema_decay": 0.0, "clipnorm": 0.5}
    model, _norm = build_leafcnn(num_classes=2, img_size=48, widths=[16, 32], drop_block=0.1, drop_top=0.3,
                                 l2_reg=1e-4, seed=1)
    model.compile(build_optimizer(cfg, 1e-3), build_loss(cfg), ["accuracy"])
    model.fit(seq, epochs=1, verbose=0)
    assert len(calls) == len(seq) - 1
    seq.close()


def test_balancer_with_the_gpu_huffman_decoder_equals_the_host_decoded_run(cuda, tmp_path, monkeypatch):
    """The balancer's input step with the Huffman decoding on the GPU (the default) against the same job with the
    codec workers doing it (LEAFFLICTION_GPU_HUFFMAN=0, the path the reference golden above pinned in round 2): the
    same files byte for byte and the same counts, on a class whose sources are a file with restart markers, one with
    optimised Huffman tables, one damaged inside its scan (the GPU hands it back, Pillow conceals the damage and the
    task goes on with Pillow's pixels, as in the reference's worker) and one cut short (Pillow raises: a failed task)."""
    import io
    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    src = tmp_path / "images"
    build_tree(src, {"Apple": {"Apple_healthy": 14}}, 64, 900)
    d = src / "Apple" / "Apple_scab"
    d.mkdir(parents=True)

    def jpeg(seed, **kw):
        b = io.BytesIO()
        Image.fromarray(leaf_like(64, 64, seed)).save(b, format="JPEG", quality=95, **kw)
        return b.getvalue()
    whole = jpeg(950)
    (d / "image (1).JPG").write_bytes(jpeg(951, restart_marker_rows=1))
    (d / "image (2).JPG").write_bytes(jpeg(952, optimize=True))
    (d / "image (3).JPG").write_bytes(whole[:len(whole) // 2] + whole[-2:])     # half the scan gone, EOI in place
    (d / "image (4).JPG").write_bytes(jpeg(953)[:-900])                           # cut short
    runs = {}
    for mode in ("1", "0"):   # "0": the codec workers also make the distortion tasks' noise planes (numpy's stream on the host)
        monkeypatch.setenv("LEAFFLICTION_GPU_HUFFMAN", mode)
        monkeypatch.setenv("LEAFFLICTION_GPU_NOISE", mode)
        work = tmp_path / f"run{mode}"
        work.mkdir()
        monkeypatch.chdir(work)
        bal = DatasetBalancer(source_dir=str(src), target_dir=str(work / "augmented"), seed=42, workers=2)
        orig = bal._images_by_class
        bal._images_by_class = lambda orig=orig: {k: sorted(v) for k, v in orig().items()}
        bal.run()
        files = {str(p.relative_to(work / "augmented")): p.read_bytes() for p in sorted((work / "augmented").rglob("*.JPG"))}
        runs[mode] = (bal.completed, bal.failed, files)
    assert runs["1"][0] == runs["0"][0] and runs["1"][1] == runs["0"][1]
    assert runs["1"][0] >= 6 and runs["1"][1] >= 1                        # the cut file's tasks failed, the others ran
    assert sorted(runs["1"][2]) == sorted(runs["0"][2])
    for name, data in runs["1"][2].items():
        assert data == runs["0"][2][name], name
    used = {Path(n).name.split("_aug_")[0] for n in runs["1"][2] if "Apple_scab" in n and "_aug_" in n}
    assert {"image (1)", "image (2)", "image (3)"} <= used, used          # the odd files really were sources
