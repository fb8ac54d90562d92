"""What a small host table costs on its way to the device (development aid): the balancer's main thread uploads a
dozen index / parameter tables per chunk."""
import time

import numpy as np
import torch

dev = torch.device("cuda:0")
vals = list(range(256))
big = torch.empty((64, 1 << 20), dtype=torch.float32, device=dev)


def busy():
    for _ in range(4):
        big.mul_(1.0001)


def timeit(name, fn, load):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(200):
        if load:
            busy()
        t0 = time.perf_counter()
        fn()
        t += time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:46s} {'GPU busy' if load else 'GPU idle'}: {t / 200 * 1e6:8.1f} us per table", flush=True)


stage = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
side = torch.cuda.Stream(dev)
arr = np.arange(256, dtype=np.int64)


def plain():
    return torch.tensor(vals, dtype=torch.int64, device=dev)


def from_numpy():
    return torch.from_numpy(arr).to(dev)


def pinned_each():
    return torch.tensor(vals, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)


def staged():
    t = torch.tensor(vals, dtype=torch.int64)
    s = stage[:t.numel() * 8].view(torch.int64)
    s.copy_(t)
    return s.to(dev, non_blocking=True)


def side_stream():
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        d = torch.tensor(vals, dtype=torch.int64).to(dev)
    cur.wait_stream(side)
    d.record_stream(cur)
    return d


def host_only():
    return torch.tensor(vals, dtype=torch.int64)


for load in (False, True):
    timeit("torch.tensor(list, device=dev)", plain, load)
    timeit("torch.from_numpy(arr).to(dev)", from_numpy, load)
    timeit("tensor.pin_memory().to(dev, non_blocking)", pinned_each, load)
    timeit("copy into one pinned buffer, .to(non_blocking)", staged, load)
    timeit("pageable copy on a side stream", side_stream, load)
    timeit("torch.tensor(list) on the host only", host_only, load)
