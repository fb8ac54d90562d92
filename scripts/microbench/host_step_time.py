"""Host time per training step against the step's wall time (is the step launch-bound?):
python scripts/microbench/host_step_time.py [bf16|f32]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from leaffliction_amd.model.cnn import LeafCNN  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    dev = torch.device("cuda:0")
    model = LeafCNN(num_classes=bench.NUM_CLASSES, img_size=bench.IMG, widths=bench.WIDTHS, drop_block=0.15,
                    drop_top=0.40, l2_reg=1e-4, augment=True, use_se=True, seed=42, device=dev)
    model.set_training_dtype(mode)
    n = bench.BATCH
    x = torch.randint(0, 256, (n, bench.IMG, bench.IMG, 3), dtype=torch.uint8).to(dev)
    y = torch.nn.functional.one_hot(torch.randint(0, bench.NUM_CLASSES, (n,)), bench.NUM_CLASSES).float().to(dev)
    for _ in range(8):
        model.train_step(x, y, 1e-3)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    host = []
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(30):
        a = time.perf_counter()
        model.train_step(x, y, 1e-3)
        host.append(time.perf_counter() - a)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: wall {1e3 * (t2 - t0) / 30:.2f} ms/step, host enqueue {1e3 * sum(host) / 30:.2f} ms/step "
          f"(median {1e3 * sorted(host)[15]:.2f}), host loop done {1e3 * (t1 - t0) / 30:.2f} ms/step")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
