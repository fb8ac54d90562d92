"""Data-parallel `fit()`: every rank must issue the same collectives whatever its shard holds.

CPU half (gloo, world size 2): `LeafCNN.fit` / `train_step` drive a stand-in model whose
forward/backward and optimizer are plain torch (the HIP kernels need a GPU) — what is under
test is the control flow the ranks share: a rank whose rank-strided slice of a ragged last
global batch is EMPTY still takes the step (zero gradient into the all-reduce, optimizer and
schedule advance), so nobody waits in a collective and the replicas end bit-equal.
GPU half (`-m gpu`, two ranks sharing the card over gloo): the same with the real model, plus
2-rank gradients against a 1-rank run at the same global batch.
"""
import os
import socket
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from leaffliction_amd.dataio.manifest import ManifestItem
from leaffliction_amd.dataio.sequence import ManifestSequence
from leaffliction_amd.model.cnn import LeafCNN


class _TorchStandIn(LeafCNN):
    """LeafCNN's training loop over a linear softmax on per-channel means (CPU)."""

    def __init__(self, num_classes: int, seed: int = 0):
        self.num_classes = num_classes
        self.device = torch.device("cpu")
        g = torch.Generator().manual_seed(seed)
        self.flat_p = torch.randn(3 * num_classes + num_classes, generator=g) * 0.1
        self.flat_g = torch.zeros_like(self.flat_p)
        self.opt_step = 0
        self.stop_training = False
        self._compiled = {}
        self._global_n = None
        self.steps_with_data = 0
        self.n_params = int(self.flat_p.numel())

    def grad_split(self):   # "late" gradients = the bias: the stand-in's backward has nothing to overlap, the
        return 3 * self.num_classes   # exchange still goes out in the two pieces train_step cuts

    def _forward_backward(self, x, y_true, between=None):
        c = self.num_classes
        x = torch.as_tensor(x, dtype=torch.float32)
        feat = x.mean((1, 2))                                   # [n,3]
        w, b = self.flat_p[:3 * c].view(3, c), self.flat_p[3 * c:]
        probs = torch.softmax(feat @ w + b, -1)
        loss = -(y_true * torch.log(probs.clamp_min(1e-7))).sum(-1)
        d = (probs - y_true) / float(self._global_n or x.shape[0])
        self.flat_g[:3 * c] = (feat.t() @ d).reshape(-1)
        self.flat_g[3 * c:] = d.sum(0)
        self.steps_with_data += 1
        if between is not None:
            between()
        return probs, loss

    def _optimizer_update(self, lr, *, weight_decay, clipnorm, ema_decay):
        self.flat_p -= lr * self.flat_g

    def l2_penalty(self):
        return torch.zeros(())

    def evaluate(self, data, verbose=0, dp=None):
        return [0.0, 0.0]


def _write_images(root: Path, n: int, size: int, dup: bool = False):
    from PIL import Image
    rng = np.random.RandomState(3)
    items = []
    for i in range(n):
        if dup and i % 2 == 1:
            arr = prev
        else:
            arr = rng.randint(0, 256, (size, size, 3)).astype(np.uint8)
        prev = arr
        cls = "a" if (i // (2 if dup else 1)) % 2 == 0 else "b"
        p = root / f"img_{i}.jpg"
        Image.fromarray(arr).save(p, quality=95)
        items.append(ManifestItem(plant="P", cls=cls, label=f"P__{cls}", split="train", src=str(p),
                                  id=f"P/{cls}/img_{i}.jpg"))
    return items


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fit_worker(rank, world, port, root, n_items, batch, out_dir):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LEAFFLICTION_DIST_TIMEOUT": "180"})
    from leaffliction_amd.train.parallel import DataParallel
    from leaffliction_amd.train.utils import CosineDecay
    dp = DataParallel(backend="gloo")
    items = _write_images(Path(root), n_items, 8) if rank == 0 else None
    dp.barrier()
    if items is None:
        items = [ManifestItem(plant="P", cls=("a" if i % 2 == 0 else "b"),
                              label="P__" + ("a" if i % 2 == 0 else "b"), split="train",
                              src=str(Path(root) / f"img_{i}.jpg"), id=f"P/x/img_{i}.jpg")
                 for i in range(n_items)]
    l2i = {"P__a": 0, "P__b": 1}
    seq = ManifestSequence(items, l2i, 8, batch, shuffle=True, seed=7, num_classes=2, one_hot=True,
                           as_numpy=True, rank=rank, world=world)
    m = _TorchStandIn(2)
    steps = len(seq) * 3
    m.compile(optimizer={"name": "adam", "schedule": CosineDecay(0.5, steps)},
              loss={"label_smoothing": 0.02})
    m.fit(seq, epochs=3, dp=dp, verbose=0)
    torch.save({"p": m.flat_p, "opt_step": m.opt_step, "with_data": m.steps_with_data},
               Path(out_dir) / f"r{rank}.pt")
    dp.shutdown()


def test_fit_ragged_last_batch_gloo_world2(tmp_path):
    """5 items, global batch 4, 2 ranks: the last global batch holds ONE item, so rank 1's slice
    is empty.  Both ranks must finish (no hang), take the same number of optimizer steps and
    end with bit-equal parameters — equal to a single-process run up to summation order."""
    root = tmp_path / "img"
    root.mkdir()
    mp.spawn(_fit_worker, args=(2, _free_port(), str(root), 5, 4, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["opt_step"] == r1["opt_step"] == 6            # 2 batches x 3 epochs on BOTH ranks
    assert r0["with_data"] == 6 and r1["with_data"] == 3    # rank 1 had nothing in 3 of them
    assert torch.equal(r0["p"], r1["p"])
    # single process, same global batches
    os.environ.update({"RANK": "0", "WORLD_SIZE": "1"})
    from leaffliction_amd.train.utils import CosineDecay
    items = [ManifestItem(plant="P", cls=("a" if i % 2 == 0 else "b"),
                          label="P__" + ("a" if i % 2 == 0 else "b"), split="train",
                          src=str(root / f"img_{i}.jpg"), id=f"P/x/img_{i}.jpg") for i in range(5)]
    seq = ManifestSequence(items, {"P__a": 0, "P__b": 1}, 8, 4, shuffle=True, seed=7, num_classes=2,
                           one_hot=True, as_numpy=True)
    m = _TorchStandIn(2)
    m.compile(optimizer={"name": "adam", "schedule": CosineDecay(0.5, 6)}, loss={"label_smoothing": 0.02})
    m.fit(seq, epochs=3, verbose=0)
    assert m.opt_step == 6
    assert torch.allclose(m.flat_p, r0["p"], rtol=1e-5, atol=1e-6)


def test_early_stopping_restores_best_without_stopping():
    """Keras 3: best weights come back in on_train_end even when patience never ran out."""
    from leaffliction_amd.train.utils import EarlyStopping

    class M:
        stop_training = False
        flat_p = torch.zeros(2)
        flat_s = torch.zeros(2)
    m = M()
    cb = EarlyStopping(patience=6, restore_best_weights=True)
    cb.set_model(m)
    cb.on_train_begin()
    for epoch, vl in enumerate([1.0, 0.5, 0.7, 0.6]):
        m.flat_p.fill_(float(epoch))
        m.flat_s.fill_(float(10 + epoch))
        cb.on_epoch_end(epoch, {"val_loss": vl})
    assert not m.stop_training and cb.best_epoch == 1
    cb.on_train_end()
    assert m.flat_p.tolist() == [1.0, 1.0] and m.flat_s.tolist() == [11.0, 11.0]
    # patience: wait counts epochs since the last improvement; no stop at epoch 0
    m2 = M()
    cb = EarlyStopping(patience=2)
    cb.set_model(m2)
    cb.on_train_begin()
    seen = []
    for epoch, vl in enumerate([0.5, 0.6, 0.7, 0.8]):
        cb.on_epoch_end(epoch, {"val_loss": vl})
        seen.append(m2.stop_training)
        if m2.stop_training:
            break
    assert seen == [False, False, True] and cb.stopped_epoch == 2
    cb1 = EarlyStopping(patience=0)                    # patience 0 cannot stop at epoch 0
    m3 = M()
    cb1.set_model(m3)
    cb1.on_train_begin()
    cb1.on_epoch_end(0, {"val_loss": float("inf")})
    assert not m3.stop_training


# ------------------------------------------------------------------ GPU: the real model
def _gpu_fit_worker(rank, world, port, root, n_items, batch, out_dir, dup, dtype="f32", bucket="f32", overlap="1"):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LEAFFLICTION_DIST_TIMEOUT": "180",
                       "LEAFFLICTION_GRAD_BUCKET": bucket, "LEAFFLICTION_GRAD_OVERLAP": overlap})
    torch.cuda.set_device(0)
    from leaffliction_amd.train.parallel import DataParallel
    from leaffliction_amd.train.utils import CosineDecay
    dp = DataParallel(backend="gloo", device=torch.device("cuda", 0))
    l2i = {"P__a": 0, "P__b": 1}
    items = [ManifestItem(plant="P", cls=c, label="P__" + c, split="train",
                          src=str(Path(root) / f"img_{i}.jpg"), id=f"P/{c}/img_{i}.jpg")
             for i in range(n_items) for c in (["a", "b"][(i // (2 if dup else 1)) % 2],)]
    seq = ManifestSequence(items, l2i, 32, batch, shuffle=not dup, seed=7, num_classes=2, one_hot=True,
                           rank=rank, world=world)
    # bf16: widths that take the mixed-precision kernels (multiples of 32), as BASELINE configs[3] runs them
    m = LeafCNN(num_classes=2, img_size=32, widths=[32, 64] if dtype == "bf16" else [16, 32], l2_reg=1e-4,
                use_norm=False, seed=5,
                drop_block=0.0 if dup else 0.15, drop_top=0.0 if dup else 0.4, augment=not dup)
    m.set_training_dtype(dtype)
    assert dp.bucket_dtype == bucket and dp.overlap == (world > 1 and overlap == "1")
    dp.broadcast_(m.flat_p, 0)
    m.reseed_step_rng(5 + rank)
    if dup:   # gradients of ONE global batch: forward/backward + all-reduce, no optimizer
        bx, by = seq[0]
        m.compile(loss={"label_smoothing": 0.02})
        yt, _ = m._targets(by)
        m._global_n = seq.global_batch_size(0)
        m._forward_backward(bx, yt)
        dp.allreduce_grads(m.flat_g)
        torch.cuda.synchronize()
        torch.save({"g": m.flat_g.cpu()}, Path(out_dir) / f"g{world}_{rank}_{dtype}_{bucket}.pt")
    else:
        m.compile(optimizer={"name": "adamw", "schedule": CosineDecay(2e-3, len(seq) * 2),
                             "weight_decay": 1e-4, "clipnorm": 0.5, "ema_decay": 0.999},
                  loss={"label_smoothing": 0.02})
        m.fit(seq, epochs=2, dp=dp, verbose=0)
        torch.cuda.synchronize()
        torch.save({"p": m.flat_p.cpu(), "ema": m.flat_ema.cpu(), "opt_step": m.opt_step,
                    "graphs": sum(st["graph"] is not None for st in m._graphs.values())},
                   Path(out_dir) / f"r{rank}.pt")
    dp.shutdown()


# (training dtype, gradient-bucket dtype): fp32 as configs[1] shards it, and BASELINE configs[3]'s combination —
# the bf16 step under data parallelism — with the exact fp32 bucket and with the halved bf16 bucket
DP_MODES = [("f32", "f32"), ("bf16", "f32"), ("bf16", "bf16")]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bucket", DP_MODES)
def test_fit_ragged_last_batch_real_model_world2(cuda, tmp_path, dtype, bucket):
    """`fit` under world size 2 with the real model: 5 items, global batch 4 (rank 1's slice of the last global
    batch is empty), 2 epochs.  Both ranks finish, take the same steps and end with BIT-equal parameters and EMA,
    whatever the step's precision and the bucket's: every rank all-reduces the same bits and applies the same
    deterministic optimizer kernels."""
    root = tmp_path / "img"
    root.mkdir()
    _write_images(root, 5, 32)
    mp.spawn(_gpu_fit_worker, args=(2, _free_port(), str(root), 5, 4, str(tmp_path), False, dtype, bucket),
             nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["opt_step"] == r1["opt_step"] == 4
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ema"], r1["ema"])
    assert torch.isfinite(r0["p"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bucket", [("bf16", "f32"), ("bf16", "bf16")])
def test_fit_graph_replay_world2_bf16(cuda, tmp_path, dtype, bucket):
    """Enough full batches per rank that the bf16 step is REPLAYED from its HIP graph under data parallelism (the
    third step of a shape is captured): 16 items, global batch 4, 2 epochs = 8 steps of 2 images per rank.
    Bit-equal replicas, and at least one captured graph on each rank."""
    root = tmp_path / "img"
    root.mkdir()
    _write_images(root, 16, 32)
    mp.spawn(_gpu_fit_worker, args=(2, _free_port(), str(root), 16, 4, str(tmp_path), False, dtype, bucket),
             nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["opt_step"] == r1["opt_step"] == 8
    assert r0["graphs"] >= 1 and r1["graphs"] >= 1
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ema"], r1["ema"])
    assert torch.isfinite(r0["p"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bucket", DP_MODES)
def test_overlapped_exchange_leaves_the_same_bits(cuda, tmp_path, dtype, bucket):
    """The data-parallel step sends the gradient bucket in two pieces, the first while stage 0's backward still
    computes (train_step's `grad_overlap`).  Same kernels in the same order on the compute
    stream, the same two element-wise sums: after 2 epochs (graph replays included) the parameters and the EMA must
    equal those of the plain step (one all-reduce after the whole backward pass, LEAFFLICTION_GRAD_OVERLAP=0)
    bit for bit, on both ranks."""
    root = tmp_path / "img"
    root.mkdir()
    _write_images(root, 16, 32)
    res = {}
    for ov in ("1", "0"):
        out = tmp_path / f"ov{ov}"
        out.mkdir()
        mp.spawn(_gpu_fit_worker, args=(2, _free_port(), str(root), 16, 4, str(out), False, dtype, bucket, ov),
                 nprocs=2, join=True)
        res[ov] = [torch.load(out / f"r{r}.pt") for r in range(2)]
        assert torch.equal(res[ov][0]["p"], res[ov][1]["p"])
    assert res["1"][0]["opt_step"] == res["0"][0]["opt_step"] == 8
    assert torch.equal(res["1"][0]["p"], res["0"][0]["p"]) and torch.equal(res["1"][0]["ema"], res["0"][0]["ema"])
    assert torch.isfinite(res["1"][0]["p"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bucket", DP_MODES)
def test_two_rank_gradients_equal_one_rank(cuda, tmp_path, dtype, bucket):
    """Global batch {a,a,b,b}: the rank-strided shards are {a,b} and {a,b}, so per-GPU BatchNorm
    statistics equal the global ones and the all-reduced 2-rank gradient must equal the 1-rank
    gradient of the whole batch — up to fp32 summation order with the fp32 bucket (also under the bf16 step, whose
    roundings are per element and therefore the same in both runs), and up to the bucket's own rounding with the
    bf16 bucket: each rank's contribution is rounded to bf16 (relative 2^-9), the sum of the two once more,
    i.e. at most 2^-8 + 2^-9 of the larger of |g| and the two contributions — 3 * 2^-9 * |g1| here, since both
    ranks contribute g1 / 2."""
    root = tmp_path / "img"
    root.mkdir()
    _write_images(root, 8, 32, dup=True)
    mp.spawn(_gpu_fit_worker, args=(2, _free_port(), str(root), 8, 8, str(tmp_path), True, dtype, bucket),
             nprocs=2, join=True)
    mp.spawn(_gpu_fit_worker, args=(1, _free_port(), str(root), 8, 8, str(tmp_path), True, dtype, bucket),
             nprocs=1, join=True)
    g2a = torch.load(tmp_path / f"g2_0_{dtype}_{bucket}.pt")["g"]
    g2b = torch.load(tmp_path / f"g2_1_{dtype}_{bucket}.pt")["g"]
    g1 = torch.load(tmp_path / f"g1_0_{dtype}_{bucket}.pt")["g"]
    assert torch.equal(g2a, g2b)
    assert g1.abs().max() > 0
    scale = g1.abs().max().item()
    if bucket == "f32":
        # bf16 step: a 2-image shard and the 4-image batch sum their statistics / weight gradients in another
        # order, and a statistic that moves by one fp32 ulp can flip the bf16 rounding of single activations
        tol = 2e-4 * scale if dtype == "f32" else 2e-2 * scale
        assert (g2a - g1).abs().max().item() <= tol, (g2a - g1).abs().max().item() / scale
    else:
        exact = (g2a - g1).abs() <= 3 * 2.0 ** -9 * g1.abs() + 2e-2 * scale
        assert bool(exact.all()), ((g2a - g1).abs().max().item() / scale)
        # the bucket really was bf16: every summed value is representable in bf16
        assert torch.equal(g2a, g2a.to(torch.bfloat16).float())
