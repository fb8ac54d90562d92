"""End-to-end `Augmentation` (DatasetBalancer) throughput on synthetic 224x224 JPEGs: host JPEG
decode -> PCIe -> HIP kernels -> PCIe -> host JPEG encode (BASELINE configs[2] shape, scaled
down).  The kernel-only rates are in bench.py's `augment` table; this is the rate a user of the
CLI sees, and it is bound by libjpeg on the host cores.
usage: python scripts/augment_e2e.py [n_majority] [n_minority] [workers]"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
from PIL import Image

sys.path.insert(0, ".")


def main():
    n_major = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    n_minor = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else len(os.sched_getaffinity(0))
    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    rng = np.random.RandomState(0)
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = Path(tmp) / "images", Path(tmp) / "augmented"
        base = [rng.randint(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(16)]
        for cls, n in (("Apple_healthy", n_major), ("Apple_rust", n_minor), ("Apple_scab", n_minor)):
            d = src / "Apple" / cls
            d.mkdir(parents=True)
            for i in range(n):
                Image.fromarray(base[i % 16]).save(d / f"image ({i + 1}).JPG", quality=95)
        os.chdir(tmp)
        bal = DatasetBalancer(source_dir=str(src), target_dir=str(dst), seed=42, workers=workers)
        bal.analyze_distribution()
        bal.calculate_plan()
        t0 = time.perf_counter()
        if os.environ.get("LF_PROFILE"):
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            bal.execute_balancing()
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
        else:
            bal.execute_balancing()
        dt = time.perf_counter() - t0
        print(f"generated {bal.completed} images ({bal.failed} failed) in {dt:.1f} s with {workers} host threads: "
              f"{bal.completed / dt:.0f} images/s end to end (decode + H2D + kernels + D2H + encode + copy of originals)")


if __name__ == "__main__":
    main()
