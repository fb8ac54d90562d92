"""Time of the GPU Huffman decoder (ops.jpeg_huffman_u8) against the host reader it replaces, on Pillow-written
224 x 224 quality-95 files (development aid).  python scripts/microbench/jpeg_huffman.py [n_images]"""
import io
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leaffliction_amd import ops  # noqa: E402
from leaffliction_amd.utils import jpeg_host  # noqa: E402


def scene(h, w, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 17.0 + seed) * np.cos(yy / 23.0), 90 + 80 * np.cos(xx / 9.0),
                    140 + 60 * np.sin((xx + yy) / 31.0)], -1) + r.normal(0, 3 + 4 * (seed % 3), (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    h = w = 224
    files = []
    for i in range(n):
        b = io.BytesIO()
        Image.fromarray(scene(h, w, i)).save(b, format="JPEG", quality=95)
        files.append(b.getvalue())
    stride = (2 * h * w * 3 + 4095) // 4096 * 4096
    slots = np.zeros((n, stride), np.uint8)
    t0 = time.perf_counter()
    for i, f in enumerate(files):
        assert jpeg_host.scan_prepare_into(f, slots[i]) is not None
    t_prep = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    ref = np.zeros((n, stride), np.uint8)
    for i, f in enumerate(files):
        assert jpeg_host.read_file_into(f, ref[i]) is not None
    t_host = (time.perf_counter() - t0) / n
    dev = torch.from_numpy(slots).cuda()
    for seq in (False, True):
        for sub in sorted({64, n}):
            x = dev[:sub].clone()
            for _ in range(3):
                st = ops.jpeg_huffman_u8(x, h, w, sequential=seq)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                st = ops.jpeg_huffman_u8(x, h, w, sequential=seq)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            assert int(st.abs().sum()) == 0
            print(f"n={sub} {'lane-per-image' if seq else 'workgroup-per-image'}: GPU Huffman {ms:.3f} ms per launch = "
                  f"{sub / ms * 1e3:.0f} images/s", flush=True)
    got = dev.clone()
    ops.jpeg_huffman_u8(got, h, w)
    if os.environ.get("LF_HUFF_STATS"):   # a build with -DLF_HUFF_STATS (LEAFHIP_LIB points at it): cycles per phase
        st = got.cpu().numpy()[:, :64].view(np.uint64).astype(np.float64)
        names = ["tables", "stage the scan", "rounds", "block numbers", "write pass", "DC scan", "n rounds", "subsequences"]
        print({k: (round(float(st[:, i].mean()), 1), float(st[:, i].max())) for i, k in enumerate(names)}, flush=True)
    m = 3 * h * w
    assert np.array_equal(got.cpu().numpy()[:, 256:256 + m], ref[:, 256:256 + m])
    # the other two codec-edge kernels of the balancer, at a chunk's size: the noise planes of its distortion tasks
    # (one in six of 256) and the encoder over rotated canvases of different sizes
    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    seeds = list(range(1000, 1043))
    ms = timed(lambda: ops.legacy_normal_u8(seeds, 0.0, 5.0, h * w * 3, dev.device))
    print(f"noise planes: {len(seeds)} x {h * w * 3} bytes in {ms:.3f} ms = {len(seeds) / ms * 1e3:.0f} planes/s "
          f"(host: 2.6 ms per plane and core)", flush=True)
    rng = np.random.RandomState(1)
    sizes = [(int(a), int(b)) for a, b in zip(rng.randint(224, 307, 43), rng.randint(224, 307, 43))]
    room = stride
    canv = torch.randint(0, 256, (len(sizes), room), dtype=torch.uint8, device=dev.device)
    items = [(i * room, a, b) for i, (a, b) in enumerate(sizes)]

    def enc():
        c = canv.clone()
        ops.jpeg_encode_items_u8(c.view(-1), items, room, 95)
    ms_clone = timed(lambda: canv.clone())
    ms = timed(enc) - ms_clone
    print(f"encoder, {len(sizes)} canvases of 224..306 pixels a side (noise: the longest scans): {ms:.3f} ms = "
          f"{len(sizes) / ms * 1e3:.0f} images/s", flush=True)
    print(f"mean file {np.mean([len(f) for f in files]):.0f} bytes; host: markers only {t_prep * 1e6:.1f} us/image, "
          f"markers + Huffman {t_host * 1e6:.1f} us/image (one core)")
