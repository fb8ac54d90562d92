"""CPU tests of the host-side mirrors: loader batching/shuffle/sharding, metrics, schedules,
and the world_size-2 gloo path of the data-parallel glue."""
import json
import os
import random
import socket
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from PIL import Image

from conftest import leaf_like
from leaffliction_amd.dataio.manifest import ManifestItem, build_label_mapping
from leaffliction_amd.dataio.sequence import ManifestSequence
from leaffliction_amd.train.parallel import shard_slice, split_counts
from leaffliction_amd.train.utils import CosineDecay, build_loss, build_optimizer
from leaffliction_amd.utils.metrics import compute_classification_metrics


def make_items(tmp_path, n, size=16):
    items = []
    for i in range(n):
        p = tmp_path / f"img_{i}.jpg"
        Image.fromarray(leaf_like(size, size, i)).save(p, quality=95)
        lab = f"P__c{i % 3}"
        items.append(ManifestItem(id=str(i), plant="P", cls=f"c{i % 3}", label=lab, split="train", src=p))
    return items


def test_sequence_batching_shuffle_and_labels(tmp_path):
    """sequence.py:47,61-72,92-99,116-119: seeded shuffle at init and per epoch, short last batch."""
    items = make_items(tmp_path, 11)
    l2i = build_label_mapping(items)
    seq = ManifestSequence(items, l2i, 16, 4, shuffle=True, seed=7, num_classes=3, one_hot=True,
                           as_numpy=True)
    rng = random.Random(7)
    idx = list(range(11))
    rng.shuffle(idx)
    assert seq.indexes == idx and len(seq) == 3
    X, y = seq[2]
    assert X.shape == (3, 16, 16, 3) and X.dtype == np.float32 and y.shape == (3, 3)
    assert 0.0 <= X.min() and X.max() <= 1.0
    assert [int(v) for v in y.argmax(-1)] == [l2i[items[i].label] for i in idx[8:]]
    seq.on_epoch_end()
    rng.shuffle(idx)
    assert seq.indexes == idx
    sparse = ManifestSequence(items, l2i, 16, 4, shuffle=False, seed=7, as_numpy=True)
    _x, ys = sparse[0]
    assert ys.dtype == np.int32 and ys.tolist() == [0, 1, 2, 0]
    with pytest.raises(ValueError):
        ManifestSequence(items, l2i, 16, 4, shuffle=False, seed=1, one_hot=True)
    assert len(list(iter(sparse))) == 3


def test_sequence_rank_sharding_partitions_every_global_batch(tmp_path):
    items = make_items(tmp_path, 10)
    l2i = build_label_mapping(items)
    seqs = [ManifestSequence(items, l2i, 16, 4, shuffle=True, seed=3, as_numpy=True, rank=r, world=3)
            for r in range(3)]
    for b in range(len(seqs[0])):
        parts = [s.batch_indexes(b) for s in seqs]
        merged = sorted(i for p in parts for i in p)
        start = b * 4
        assert merged == sorted(seqs[0].indexes[start:start + 4])
        assert sum(len(p) for p in parts) == seqs[0].global_batch_size(b)
    assert shard_slice(list(range(7)), 1, 3) == [1, 4]
    assert split_counts(7, 3) == (3, 2, 2)


def test_cosine_and_optimizer_config():
    sched = CosineDecay(2e-3, 100)
    assert sched(0) == pytest.approx(2e-3) and sched(50) == pytest.approx(1e-3)
    assert sched(100) == pytest.approx(0.0, abs=1e-12) and sched(1000) == pytest.approx(0.0, abs=1e-12)
    reg = {"optimizer": "adamw", "lr": 0.002, "weight_decay": 1e-4, "label_smoothing": 0.02,
           "clipnorm": 0.5, "ema_decay": 0.999}
    opt = build_optimizer(reg, sched)
    assert opt["name"] == "adamw" and opt["weight_decay"] == 1e-4 and opt["clipnorm"] == 0.5
    fast = build_optimizer({"optimizer": "adam", "weight_decay": 0.0, "clipnorm": 0.0}, 3e-3)
    assert fast["name"] == "adam" and fast["weight_decay"] == 0.0 and fast["lr"] == 3e-3
    assert build_loss(reg)["label_smoothing"] == 0.02
    assert build_loss({"label_smoothing": 0.0})["name"] == "sparse_categorical_crossentropy"


def test_metrics_match_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    rng = np.random.RandomState(0)
    yt, yp = rng.randint(0, 4, 300).tolist(), rng.randint(0, 4, 300).tolist()
    labels = ["a", "b", "c", "d"]
    m = compute_classification_metrics(yt, yp, labels)
    assert m["accuracy"] == pytest.approx(sk.accuracy_score(yt, yp))
    for avg in ("macro", "weighted"):
        assert m[f"{avg}_f1"] == pytest.approx(sk.f1_score(yt, yp, average=avg, zero_division=0))
        assert m[f"{avg}_precision"] == pytest.approx(sk.precision_score(yt, yp, average=avg, zero_division=0))
        assert m[f"{avg}_recall"] == pytest.approx(sk.recall_score(yt, yp, average=avg, zero_division=0))
    per = sk.f1_score(yt, yp, average=None, zero_division=0)
    assert [m[f"f1_{x}"] for x in labels] == pytest.approx(per.tolist())
    b = compute_classification_metrics([0, 1, 1, 0], [0, 1, 0, 0], ["n", "p"])
    assert b["binary_recall"] == pytest.approx(0.5) and b["binary_precision"] == pytest.approx(1.0)


# ----------------------------------------------------------------- gloo, world_size 2
def _dp_worker(rank, world, port, out_dir):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LEAFFLICTION_DIST_TIMEOUT": "180"})
    from leaffliction_amd.train.parallel import DataParallel
    dp = DataParallel(backend="gloo")
    assert dp.active and dp.world == world
    # the one exchange step: SUM of flat gradient buckets scaled by 1/global_n
    g_full = torch.arange(12, dtype=torch.float32).reshape(6, 2)       # per-sample "gradients"
    mine = g_full[rank::world].sum(0) / 6.0
    flat = mine.clone()
    dp.allreduce_grads(flat)
    assert torch.allclose(flat, g_full.mean(0))
    # replicas start identical
    p = torch.full((5,), float(rank))
    dp.broadcast_(p, 0)
    assert (p == 0).all()
    # integer confusion counts: exact
    cm = torch.tensor([[rank + 1, 0], [2, rank]], dtype=torch.int64)
    dp.allreduce_counts(cm)
    assert cm.tolist() == [[3, 0], [4, 1]]
    assert dp.allreduce_scalars([1.5, float(rank)]) == [3.0, 1.0]
    dp.barrier()
    Path(out_dir, f"ok_{rank}").write_text("ok")
    dp.shutdown()


def test_data_parallel_glue_gloo_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_balancer_task_list_matches_reference(tmp_path):
    """build_tasks consumes the global `random` stream exactly like the reference loop."""
    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    gold = json.loads((Path(__file__).parent / "golden" / "balancer_golden.json").read_text())
    root = tmp_path / "images"
    for plant, classes in gold["layout"].items():
        for cls, n in classes.items():
            d = root / plant / cls
            d.mkdir(parents=True)
            for i in range(n):
                (d / f"image ({i + 1}).JPG").write_bytes(b"x")
    bal = DatasetBalancer(source_dir=str(root), target_dir=str(tmp_path / "aug"), seed=gold["seed"],
                          workers=1)
    bal.analyze_distribution()
    assert bal.calculate_plan() == gold["plan"]
    images = {cls: sorted((root / "Apple" / cls).glob("*.JPG")) for cls in gold["layout"]["Apple"]}
    tasks = bal.build_tasks(images)
    assert [(Path(t["source_img"]).name, Path(t["output_path"]).name, t["transform_name"], t["seed"])
            for t in tasks] == [(t["source"], t["output"], t["transform"], t["seed"])
                                for t in gold["tasks"]]


def test_split_matches_reference_golden(tmp_path):
    """leaffliction_amd.cli.split vs the reference's split functions (tests/golden/
    make_golden_split.py): scan order, both allocation strategies, seeded split maps, summary
    rows, manifest schema, and the CLI end to end."""
    import csv
    import json
    from pathlib import Path
    from leaffliction_amd.cli import split as S
    from leaffliction_amd.dataio.manifest import load_manifest, select_items
    gold = json.loads((Path(__file__).parent / "golden" / "split_golden.json").read_text())
    root = tmp_path / "images"
    for plant, classes in gold["layout"].items():
        for cls, n in classes.items():
            d = root / plant / cls
            d.mkdir(parents=True)
            for i in range(n):
                (d / f"image ({i + 1}).{'JPG' if i % 2 else 'jpg'}").write_bytes(b"")
            (d / "notes.txt").write_text("x")
            (d / "picture.png").write_bytes(b"")
    images = S.scan_dataset(root)
    assert [im.rel_id for im in images] == gold["scan"]
    by_label = {}
    for im in images:
        by_label.setdefault(im.label, []).append(im)
    counts = {k: len(v) for k, v in by_label.items()}
    assert counts == gold["counts"]
    allocs = {"ratio_0.2": S.allocate_validation_by_ratio(counts, 0.2),
              "ratio_0.5": S.allocate_validation_by_ratio(counts, 0.5),
              "min_val_6": S.allocate_validation_counts(counts, 6),
              "min_val_100": S.allocate_validation_counts(counts, 100),
              "min_val_0": S.allocate_validation_counts(counts, 0)}
    for case in gold["cases"]:
        alloc = allocs[case["name"]]
        assert alloc == case["alloc"], case["name"]
        sm = S.build_split_map(by_label, alloc, case["seed"])
        assert sm == case["split"], (case["name"], case["seed"])
        assert [[str(v) for v in row] for row in S.summary_rows(by_label, sm)] == case["summary"]
    # CLI: defaults are ratio 0.2 / seed 32 -> the first golden case
    out = tmp_path / "out"
    S.main(["--src", str(root), "--out", str(out), "--out-manifest", str(out / "manifest_split.json")])
    doc = json.loads((out / "manifest_split.json").read_text())
    assert sorted(doc["meta"]) == gold["manifest_meta_keys"] and doc["meta"]["strategy"] == gold["manifest_strategy"]
    assert list(doc["items"][0]) == gold["manifest_item_keys"] and doc["meta"]["min_val"] == 20
    assert {it["id"]: it["split"] for it in doc["items"]} == gold["cases"][0]["split"]
    with (out / "split_summary.csv").open() as f:
        assert list(csv.reader(f)) == gold["cases"][0]["summary"]
    items = load_manifest(out / "manifest_split.json")      # and it feeds the train entrypoint's loader
    assert len(select_items(items, "val")) == 5 and len(select_items(items, "train")) == 23
    import pytest
    with pytest.raises(SystemExit) as e:
        S.main(["--src", str(tmp_path / "missing")])
    assert e.value.code == 1
    with pytest.raises(ValueError):
        S.allocate_validation_by_ratio(counts, 1.0)


def test_distribution_matches_reference_golden(tmp_path, monkeypatch):
    """leaffliction_amd.cli.Distribution / utils.distribution vs the reference's functions."""
    import csv
    import json
    from pathlib import Path
    from leaffliction_amd.cli import Distribution as D
    from leaffliction_amd.utils.distribution import count_images, merge_csv
    gold = json.loads((Path(__file__).parent / "golden" / "split_golden.json").read_text())
    root = tmp_path / "images"
    for plant, classes in gold["layout"].items():
        for cls, n in classes.items():
            d = root / plant / cls
            d.mkdir(parents=True)
            for i in range(n):
                (d / f"image ({i + 1}).{'JPG' if i % 2 else 'jpg'}").write_bytes(b"")
            (d / "notes.txt").write_text("x")
    g = gold["distribution"]
    assert [list(r) for r in count_images(root, None)] == g["rows_all"]
    assert [list(r) for r in count_images(root, {"Grape"})] == g["rows_grape"]
    p = tmp_path / "d.csv"
    p.write_text("plant,class,count\nApple,healthy,99\nPear,ripe,4\n")
    merge_csv(count_images(root, {"Grape"}), p)
    with p.open() as f:
        assert list(csv.reader(f)) == g["merged"]
    p2 = tmp_path / "e.csv"
    p2.write_text("a,b\n1,2\n")
    merge_csv(count_images(root, None), p2)
    with p2.open() as f:
        assert list(csv.reader(f)) == g["fresh"]
    monkeypatch.chdir(tmp_path)
    D.main([str(root), "--no-plots"])
    with (tmp_path / "artifacts/plots/distribution.csv").open() as f:
        assert list(csv.reader(f)) == g["fresh"]
    D.main([str(root), "--plants", "Nope", "--no-plots"])       # unknown plant: logs, returns
    D.main([str(tmp_path / "missing"), "--no-plots"])           # missing root: logs, returns


def test_worker_counts_follow_the_granted_cores(monkeypatch, tmp_path):
    """get_available_cores: affinity mask, capped by the cgroup CPU quota, shared among the ranks of a node; the
    optimal count keeps the reference's heuristic on top of it (system_info.py:9-46)."""
    import builtins
    import io
    import os

    from leaffliction_amd.utils import system_info as si
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if str(path) == "/sys/fs/cgroup/cpu.max":
            return io.StringIO(quota[0])
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    quota = ["max 100000\n"]
    assert si.get_available_cores() == 64 and si.get_optimal_worker_count() == 48
    quota[0] = "1600000 100000\n"
    assert si.get_available_cores() == 16 and si.get_optimal_worker_count() == 12
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert si.get_available_cores() == 2 and si.get_optimal_worker_count() == 1
    quota[0] = "garbage"
    assert si.get_available_cores() == 8
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    quota[0] = "350000 100000\n"
    assert si.get_available_cores() == 3 and si.get_optimal_worker_count() == 2


def test_manifest_text_equals_json_dumps_indent2():
    """The augmented manifest is written as `json.dumps(manifest, indent=2, ensure_ascii=False)` would write it
    (the reference's save_manifest), through the C encoder."""
    import json

    from leaffliction_amd.preprocessing.dataset_components import ManifestGenerator as M
    items = [{"plant": "Apple", "class": "Apple_scab é", "label": "Apple__x", "split": "train",
              "src": '/a/b "q"/c.JPG', "id": "a\\b\tc", "augmented": i % 2 == 0, "n": i, "f": 0.5, "z": None}
             for i in range(7)]
    cases = [{"meta": {"a": 1, "b": None, "c": "x", "d": {"n": [1, 2]}}, "items": items},
             {"meta": {}, "items": items}, {"meta": {"a": 1}, "items": []}, {"items": items},
             {"meta": {"a": 1}, "items": [{"x": [1]}]}, {"meta": {"a": 1}, "items": items, "z": 1},
             {"meta": {"a": 1}, "items": [{}]}, [1, 2], {"meta": {"a": 1}, "items": [{1: "a"}]}]
    for man in cases:
        assert M._dumps(man) == json.dumps(man, indent=2, ensure_ascii=False)


def test_codec_pool_job_splitting():
    """Decode / encode jobs are cut into `pieces_per_worker` runs per worker: contiguous, in order, nothing lost;
    a small batch with one piece per worker does not fall apart into one-task jobs."""
    from leaffliction_amd.preprocessing.codec_pool import CodecPool
    pool = CodecPool.__new__(CodecPool)
    pool.workers = 12
    jobs = list(range(256))
    for pieces in (1, 2, 4):
        parts = pool._split(jobs, pieces)
        assert [j for p in parts for j in p] == jobs
        assert len(parts) <= pieces * pool.workers and max(map(len, parts)) == -(-256 // (pieces * 12))
    assert [len(p) for p in pool._split(list(range(32)), 1)] == [3] * 10 + [2]
    assert len(pool._split(list(range(32)), 4)) == 32
    assert pool._split([], 2) == []


def test_fast_manifest_writer_writes_the_same_bytes(tmp_path):
    """ManifestGenerator.write_augmented_manifest == save_manifest(generate_augmented_manifest()) byte for byte (the
    reference's json.dumps(manifest, indent=2, ensure_ascii=False), dataset_components.py), on names that need
    escaping, non-ASCII names, dotted names and an `_aug_` in the extension only."""
    import re
    from leaffliction_amd.preprocessing.dataset_components import ManifestGenerator
    root = tmp_path / "aug"
    names = ["image (1).JPG", "image (1)_aug_flip_1.JPG", 'we"ird\\name.jpg', "feuille_é_ü.JPG", "a.b.c_aug_x.png",
             "noext_aug_", "plain._aug_", ".hidden_aug_.JPG"]
    for plant, cls in (("Apple", "Apple_healthy"), ("Gr\u00e4pe", 'cl"ass')):
        d = root / plant / cls
        d.mkdir(parents=True)
        for n in names:
            (d / n).write_bytes(b"x")
    (root / "stray.txt").write_text("not a plant directory")
    gen = ManifestGenerator({"meta": {"created_at": "2024-01-01T00:00:00", "seed": 7}}, tmp_path / "images", root, 3)
    a, b = tmp_path / "a.json", tmp_path / "b.json"
    gen.save_manifest(gen.generate_augmented_manifest(), a)
    assert gen.write_augmented_manifest(b) == 2 * len(names)
    stamp = re.compile(r'"augmented_at": "[^"]*"')
    ta, tb = stamp.sub("", a.read_text(encoding="utf-8")), stamp.sub("", b.read_text(encoding="utf-8"))
    assert ta == tb
    import json
    assert json.loads(b.read_text(encoding="utf-8"))["meta"]["augmented_images"] == json.loads(a.read_text(encoding="utf-8"))["meta"]["augmented_images"]
    # an empty tree
    empty = tmp_path / "empty"
    (empty / "P" / "c").mkdir(parents=True)
    g2 = ManifestGenerator({}, tmp_path, empty, 1)
    g2.save_manifest(g2.generate_augmented_manifest(), a)
    assert g2.write_augmented_manifest(b) == 0
    assert stamp.sub("", a.read_text()) == stamp.sub("", b.read_text())


def test_codec_workers_are_the_ranks_share_of_the_node(monkeypatch):
    """One process per GPU on a node: every rank starts its own codec workers, so the ceiling is the node's cores
    divided among the ranks on it (LOCAL_WORLD_SIZE under torch.distributed.run; utils.system_info)."""
    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    from leaffliction_amd.utils.system_info import get_available_cores
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    cores = get_available_cores()
    assert DatasetBalancer._host_threads(10 ** 6) == cores
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert get_available_cores() == max(1, cores // 4)
    assert DatasetBalancer._host_threads(10 ** 6) == max(1, cores // 4)
    assert DatasetBalancer._host_threads(1) == 1
