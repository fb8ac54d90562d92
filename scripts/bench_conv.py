"""Per-layer conv timing on the GPU (development aid; bench.py is the contract benchmark)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from leaffliction_amd import nn  # noqa: E402

LAYERS = [  # name, cin, cout, hw, k
    ("stem 3->32 @224", 3, 32, 224, 3),
    ("s1 32->32 @224", 32, 32, 224, 3),
    ("s2 32->64 @112", 32, 64, 112, 3),
    ("s2 64->64 @112", 64, 64, 112, 3),
    ("s3 64->128 @56", 64, 128, 56, 3),
    ("s3 128->128 @56", 128, 128, 56, 3),
    ("s4 128->256 @28", 128, 256, 28, 3),
    ("s4 256->256 @28", 256, 256, 28, 3),
    ("proj 32->64 @112", 32, 64, 112, 1),
    ("proj 128->256 @28", 128, 256, 28, 1),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    print(f"batch {n}")
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for name, cin, cout, hw, k in LAYERS:
        x = torch.randn(n, cin, hw, hw, device=dev)
        w = torch.randn(cin, k * k, cout, device=dev) * 0.05
        dy = torch.randn(n, cout, hw, hw, device=dev)
        wt = nn.conv2d_dgrad_weights(w, k)
        y = torch.empty(n, cout, hw, hw, device=dev)
        dx = torch.empty(n, cin, hw, hw, device=dev)
        flop = 2.0 * n * hw * hw * cin * cout * k * k
        t_f = timeit(lambda: nn.conv2d(x, w, k, out=y))
        t_d = timeit(lambda: nn.conv2d(dy, wt, k, out=dx))
        t_w = timeit(lambda: nn.conv2d_wgrad(x, dy, k))
        tot["fwd"] += t_f
        tot["dgrad"] += t_d
        tot["wgrad"] += t_w
        print(f"{name:20s} fwd {t_f*1e3:7.3f} ms {flop/t_f/1e12:6.1f} TF | dgrad {t_d*1e3:7.3f} ms "
              f"{flop/t_d/1e12:6.1f} TF | wgrad {t_w*1e3:7.3f} ms {flop/t_w/1e12:6.1f} TF", flush=True)
        del x, w, dy, wt, y, dx
    print("sum ms:", {k: round(v * 1e3, 2) for k, v in tot.items()})


if __name__ == "__main__":
    main()
