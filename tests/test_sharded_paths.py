"""The two paths that shard WITHOUT an exchange step: dataset balancing (one
contiguous share of the task list per GPU, rank-0 manifest) and batch inference (one contiguous
share of the file list per replica, results back in input order, integer confusion counts
summed).  World size 2 over gloo must reproduce the single-process run byte for byte.

CPU half: the balancer's pixel stage is routed to the oracle (tests may; the product never does)
and the predictor is a stand-in — what is under test is partitioning, ordering, counters, who
writes what.  GPU half (`-m gpu`): the same with the HIP kernels and the real model, two ranks
sharing the card.
"""
import hashlib
import json
import os
import shutil
import socket
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp
from PIL import Image

from conftest import leaf_like
from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
from leaffliction_amd.utils import ranks as R

LAYOUT = {"Apple": {"healthy": 9, "rust": 4, "scab": 2}, "Grape": {"esca": 3, "spot": 7}}


def test_contiguous_share_covers_everything_once():
    for total in (0, 1, 5, 8, 17, 100):
        for world in (1, 2, 3, 8):
            spans = [R.contiguous_share(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


class OracleBalancer(DatasetBalancer):
    """DatasetBalancer with the pixel stage on the CPU oracle (test infrastructure only)."""

    def _run_group(self, op, images, params):
        from oracle import pil_ops as P
        out = []
        for img, p in zip(images, params):
            if op == "flip":
                out.append(P.flip(img, p["mode"]))
            elif op == "rotate":
                out.append(P.rotate_expand_white(img, p["angle"]))
            elif op in ("skew", "shear"):
                out.append(P.warp_bicubic(img, p["coeffs"], perspective=(op == "skew")))
            elif op == "crop":
                out.append(P.crop_resize_lanczos(img, *p["box"]))
            elif "noise8" in p:   # the codec workers cast the noise to uint8 (numpy's astype): bytewise add
                out.append(P.autocontrast((img + p["noise8"]).astype(np.uint8), p["cutoff"]))
            else:
                out.append(P.autocontrast(P.noise_wrap_add(img, p["noise"]), p["cutoff"]))
        return out

    def _images_by_class(self):   # fixed directory order
        return {k: sorted(v) for k, v in sorted(super()._images_by_class().items())}


class GpuBalancer(DatasetBalancer):
    def _images_by_class(self):
        return {k: sorted(v) for k, v in sorted(super()._images_by_class().items())}


def build_tree(root: Path, size: int):
    k = 0
    for plant, classes in LAYOUT.items():
        for cls, n in classes.items():
            d = root / plant / cls
            d.mkdir(parents=True)
            for i in range(n):
                Image.fromarray(leaf_like(size, size, 900 + k)).save(d / f"image ({i + 1}).JPG", quality=95)
                k += 1


def tree_digest(root: Path):
    return {str(p.relative_to(root)): hashlib.sha256(p.read_bytes()).hexdigest()
            for p in sorted(root.rglob("*")) if p.is_file()}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _join(rank, world, port, gpu):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LEAFFLICTION_DIST_TIMEOUT": "180",
                       "LEAFFLICTION_DIST_BACKEND": "gloo"})
    if not gpu:
        os.environ["CUDA_VISIBLE_DEVICES"] = ""
    return R.init_from_env()


_KEEP_OPEN = []


def _balance_worker(rank, world, port, src, dst, work, gpu):
    import faulthandler

    import torch.distributed as dist
    # a rank that is still here after 100 s leaves its stacks behind (see _spawn_bounded)
    os.environ["LEAFFLICTION_DEBUG_STACKS"] = work     # the codec workers leave theirs as well (codec_pool._warm)
    stacks = open(os.path.join(work, f"stacks_{world}_{rank}.txt"), "w")
    _KEEP_OPEN.append(stacks)   # the watchdog writes to this descriptor after the function has returned
    faulthandler.dump_traceback_later(100, exit=False, file=stacks)
    rk = _join(rank, world, port, gpu)
    os.chdir(work)
    cls = GpuBalancer if gpu else OracleBalancer
    bal = cls(source_dir=src, target_dir=dst, seed=42, workers=2)
    assert bal.ranks.world == world
    bal.run()
    b, e = R.contiguous_share(len(bal.tasks), rank, world)
    Path(work, f"counts_{world}_{rank}.json").write_text(json.dumps(
        {"completed": bal.completed, "failed": bal.failed, "tasks": len(bal.tasks), "share": [b, e]}))
    if rk.active:
        dist.destroy_process_group()
    # the watchdog stays armed on purpose: the hang this guards against sits in the interpreter's EXIT path, after
    # this function has returned (a process that exits in time takes the watchdog thread with it)
    if world == 1 and not gpu:
        faulthandler.cancel_dump_traceback_later()   # in-process call (the pytest process itself)


def _spawn_bounded(fn, args, nprocs, limit=150.0):
    """mp.spawn with a deadline: True when every rank finished, False when the ranks had to be killed.  (Round 2 saw
    about one run in twenty of the two-rank CPU job end with both ranks asleep on a process-shared semaphore on their
    way OUT; CodecPool.close() now joins its executor itself instead of leaving that to the interpreter's exit handlers.
    A run that overruns FAILS the test — no second attempt — and the failure message carries the Python stacks each
    rank wrote after 100 s (`faulthandler`), which is the evidence to work from.)"""
    import time
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.time() + limit
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            for proc in ctx.processes:
                proc.join(10)
            return False
    return True


def _check_balancer(tmp_path, gpu):
    src = tmp_path / "images"
    build_tree(src, 48)
    runs = {}
    for world in (1, 2):
        work = tmp_path / f"w{world}"
        work.mkdir()
        dst = work / "augmented"
        if world == 1:
            _balance_worker(0, 1, 0, str(src), str(dst), str(work), gpu) if not gpu else \
                mp.spawn(_balance_worker, args=(1, _free_port(), str(src), str(dst), str(work), gpu), nprocs=1)
        else:
            if not _spawn_bounded(_balance_worker, (2, _free_port(), str(src), str(dst), str(work), gpu), 2):
                stuck = "\n".join(f"--- {f.name}\n{f.read_text()}" for f in sorted([*work.glob("stacks_2_*.txt"),
                                                                                         *work.glob("codec_*.txt")]))
                pytest.fail(f"the two-rank balancer run did not finish within its deadline; stacks of the stuck "
                            f"ranks:\n{stuck}")
        man = json.loads((work / "artifacts/datasets/manifest_augmented.json").read_text())
        man["meta"].pop("augmented_at")
        for it in man["items"]:
            it["src"] = str(Path(it["src"]).relative_to(dst))
        man["meta"]["src_root"] = ""
        man["items"].sort(key=lambda it: it["id"])
        runs[world] = (tree_digest(dst), man,
                       [json.loads((work / f"counts_{world}_{r}.json").read_text()) for r in range(world)])
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    d1, m1, c1 = runs[1]
    d2, m2, c2 = runs[2]
    n_tasks = c1[0]["tasks"]
    assert n_tasks == sum(max(cl.values()) - n for cl in LAYOUT.values() for n in cl.values())  # per plant
    assert c1[0]["completed"] == n_tasks and c1[0]["failed"] == 0
    # both ranks report the job totals; their shares tile the task list
    assert all(c["completed"] == n_tasks and c["failed"] == 0 for c in c2)
    assert c2[0]["share"][0] == 0 and c2[0]["share"][1] == c2[1]["share"][0] and c2[1]["share"][1] == n_tasks
    assert d1 == d2, "2-rank tree differs from the 1-rank tree"
    assert m1 == m2
    assert m1["meta"]["augmented_images"] == n_tasks
    return d1


def test_balancer_two_ranks_equal_one_rank_gloo(tmp_path):
    _check_balancer(tmp_path, gpu=False)


class _StubLoader:
    labels = ["P__a", "P__b", "P__c"]
    img_size = 8


class StubPredictor:
    """predict_batch = a deterministic function of the file's bytes (no model, no GPU)."""
    model_loader = _StubLoader()

    def predict_batch(self, paths):
        out = []
        for p in paths:
            h = hashlib.sha256(Path(p).read_bytes()).digest()
            pr = np.array([h[0] + 1.0, h[1] + 1.0, h[2] + 1.0])
            pr /= pr.sum()
            top = int(np.argmax(pr))
            out.append({"image_path": Path(p), "top_prediction": self.model_loader.labels[top],
                        "confidence": float(pr[top]),
                        "all_probabilities": {l: float(v) for l, v in zip(self.model_loader.labels, pr)},
                        "original_array": np.zeros(1), "processed_array": np.zeros(1)})
        return out

    from leaffliction_amd.predict.predictor import Predictor as _P
    predict_batch_sharded = _P.predict_batch_sharded


def _predict_worker(rank, world, port, files, labels, out):
    import torch.distributed as dist
    from leaffliction_amd.predict.evaluation import sharded_confusion_counts
    rk = _join(rank, world, port, False)
    pred = StubPredictor()
    res = pred.predict_batch_sharded(files, rk)
    cm = sharded_confusion_counts(pred, [Path(f) for f in files], labels, rk)
    Path(out, f"pred_{world}_{rank}.json").write_text(json.dumps(
        {"results": [{k: (str(v) if k == "image_path" else v) for k, v in r.items() if not k.endswith("_array")}
                     for r in res], "cm": cm}))
    if rk.active:
        dist.destroy_process_group()


def test_batch_inference_two_replicas_equal_one(tmp_path):
    files = []
    for i in range(11):
        p = tmp_path / f"f{i}.jpg"
        p.write_bytes(bytes([i, 7 * i % 256, 255 - i]) * 5)
        files.append(str(p))
    labels = [_StubLoader.labels[i % 3] for i in range(11)]
    _predict_worker(0, 1, 0, files, labels, str(tmp_path))
    mp.spawn(_predict_worker, args=(2, _free_port(), files, labels, str(tmp_path)), nprocs=2)
    os.environ.pop("WORLD_SIZE", None)
    one = json.loads((tmp_path / "pred_1_0.json").read_text())
    two = [json.loads((tmp_path / f"pred_2_{r}.json").read_text()) for r in range(2)]
    assert [r["image_path"] for r in one["results"]] == files          # input order
    assert two[0] == one and two[1] == one
    assert sum(map(sum, one["cm"])) == 11


# ------------------------------------------------------------------ GPU: kernels + real model
@pytest.mark.gpu
def test_balancer_two_ranks_equal_one_rank_gpu(cuda, tmp_path):
    digest = _check_balancer(tmp_path, gpu=True)
    # and the GPU tree equals the oracle's (CPU) tree, file for file
    cpu = tmp_path / "cpu"
    cpu.mkdir()
    _balance_worker(0, 1, 0, str(tmp_path / "images"), str(cpu / "augmented"), str(cpu), False)
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    assert tree_digest(cpu / "augmented") == digest


# ------------------------------------------------------------------ failure paths of the job's preparation
def test_a_failed_copy_of_the_originals_fails_the_job(tmp_path, monkeypatch):
    """The originals' bytes follow behind the pipeline (copied by the codec worker processes); when that copy fails
    (disk full, a source file gone) the job must raise as the reference's copytree would (dataset_balancer.py:70-81),
    not report success over a tree of zero-byte placeholders.  Here one source entry is a link to nothing: its
    placeholder can be laid out, its bytes cannot be copied — the failure happens in a worker process and has to
    travel back to the thread that raises it."""
    src = tmp_path / "images"
    build_tree(src, 16)
    leaf = next(p for p in sorted(src.rglob("*")) if p.is_dir() and not any(c.is_dir() for c in p.iterdir()))
    os.symlink(str(tmp_path / "nowhere.JPG"), str(leaf / "image (99).JPG"))
    bal = OracleBalancer(source_dir=str(src), target_dir=str(tmp_path / "aug"), seed=1, workers=1)
    os.chdir(tmp_path)
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    try:
        with pytest.raises(OSError, match=r"image \(99\)"):
            bal.run()
    finally:
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    assert bal._codec is None
    assert not (tmp_path / "artifacts/datasets/manifest_augmented.json").exists()   # no manifest of a broken tree


def test_the_copy_of_the_originals_survives_a_pool_that_is_gone(tmp_path):
    """Batches the codec pool does not take (it was shut down under the copying thread, as after a failed pipeline)
    are copied by the thread itself: the tree is complete."""
    from leaffliction_amd.preprocessing.codec_pool import CodecPool
    src = tmp_path / "images"
    build_tree(src, 16)
    bal = OracleBalancer(source_dir=str(src), target_dir=str(tmp_path / "aug"), seed=1, workers=1)
    bal._codec = CodecPool(1)
    bal._codec.close()
    bal._fresh_target()
    bal._join_copy()
    for p in src.rglob("*.JPG"):
        assert (tmp_path / "aug" / p.relative_to(src)).read_bytes() == p.read_bytes()


def _broken_prep_worker(rank, world, port, src, dst, work):
    import faulthandler
    import time

    import torch.distributed as dist
    stacks = open(os.path.join(work, f"stacks_{world}_{rank}.txt"), "w")
    _KEEP_OPEN.append(stacks)
    faulthandler.dump_traceback_later(60, exit=False, file=stacks)
    rk = _join(rank, world, port, False)
    os.chdir(work)
    bal = OracleBalancer(source_dir=src, target_dir=dst, seed=42, workers=1)
    bal.analyze_distribution()
    bal.calculate_plan()
    if rank == 0:   # the tree vanishes between planning and execution: rank 0's preparation raises
        bal.source_dir = Path(src + "_gone")
    t0 = time.time()
    try:
        bal.execute_balancing()
        outcome = "finished"
    except FileNotFoundError as e:
        outcome = f"FileNotFoundError: {e}"
    except RuntimeError as e:
        outcome = f"RuntimeError: {e}"
    Path(work, f"outcome_{rank}.json").write_text(json.dumps({"outcome": outcome, "seconds": time.time() - t0,
                                                              "codec": bal._codec is None}))
    if rk.active:
        dist.destroy_process_group()


def test_rank0_preparation_failure_reaches_every_rank(tmp_path):
    """Rank 0 prepares the job alone (fresh target tree, task list).  If that raises, the other ranks used to wait
    in the task-list broadcast until the process group timed out (30 minutes by default) with their codec workers
    running; now the failure is broadcast in the task list's place and every rank raises at once."""
    src = tmp_path / "images"
    build_tree(src, 16)
    work = tmp_path / "w"
    work.mkdir()
    ok = _spawn_bounded(_broken_prep_worker, (2, _free_port(), str(src), str(work / "augmented"), str(work)), 2,
                        limit=90.0)
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    assert ok, "\n".join(f.read_text() for f in sorted(work.glob("stacks_2_*.txt")))
    out = [json.loads((work / f"outcome_{r}.json").read_text()) for r in range(2)]
    assert out[0]["outcome"].startswith("FileNotFoundError") and "Source directory not found" in out[0]["outcome"]
    assert out[1]["outcome"].startswith("RuntimeError: rank 0 could not prepare") and "FileNotFoundError" in out[1]["outcome"]
    assert all(o["codec"] and o["seconds"] < 30 for o in out)
