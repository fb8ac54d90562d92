"""Histogram / distortion rates on SURVEY §8d's two synthetic sets: U (uniform bytes) and L
(leaf-like: flat background, one green disc, brown spots, N(0,8) noise).  Smooth images put many
lanes of a wave on the same histogram bin, which uniform noise never shows (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaffliction_amd import ops  # noqa: E402


def leaf_like_batch(n, size, dev, seed=42):
    g = torch.Generator(device=dev).manual_seed(seed)
    img = torch.normal(150.0, 8.0, (n, size, size, 1), generator=g, device=dev).expand(-1, -1, -1, 3).clone()
    yy, xx = torch.meshgrid(torch.arange(size, device=dev), torch.arange(size, device=dev), indexing="ij")
    c = torch.randint(80, 144, (n, 2), generator=g, device=dev)
    r = torch.randint(50, 90, (n,), generator=g, device=dev)
    disc = ((yy[None] - c[:, 0, None, None]) ** 2 + (xx[None] - c[:, 1, None, None]) ** 2) <= (r * r)[:, None, None]
    img[disc] = torch.tensor([60.0, 140.0, 50.0], device=dev)
    for _ in range(3):
        b = torch.randint(0, size, (n, 2), generator=g, device=dev)
        br = torch.randint(3, 11, (n,), generator=g, device=dev)
        spot = ((yy[None] - b[:, 0, None, None]) ** 2 + (xx[None] - b[:, 1, None, None]) ** 2) <= (br * br)[:, None, None]
        img[spot] = torch.tensor([120.0, 70.0, 30.0], device=dev)
    img = img + torch.normal(0.0, 8.0, img.shape, generator=g, device=dev)
    return img.clamp_(0, 255).to(torch.uint8)


def rate(fn, n, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return n / (e0.elapsed_time(e1) * 1e-3 / iters)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda:0")
    sets = {"U": torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device=dev),
            "L": leaf_like_batch(n, 224, dev)}
    cut = torch.rand(n, dtype=torch.float64, device=dev) * 2
    for name, x in sets.items():
        img_b = 224 * 224 * 3
        r_hist = rate(lambda: ops.hist_u8(x), n)
        r_stats = rate(lambda: ops.hsv_region_stats(x), n)
        r_dist = rate(lambda: ops.autocontrast_u8(ops.noise_philox_add_u8(x, 42, 5.0), cut), n)
        print(f"set {name}: hist {r_hist / 1e6:.2f} M img/s ({r_hist * (img_b + 3072) / 1e12:.2f} TB/s)  "
              f"hsv_stats {r_stats / 1e6:.2f} M img/s  distortion {r_dist / 1e6:.2f} M img/s", flush=True)
