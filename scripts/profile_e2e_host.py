"""Where the main process spends its time in DatasetBalancer.execute_balancing (cProfile; development aid)."""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    dev = torch.device("cuda:0")
    tmp = Path(tempfile.mkdtemp(prefix="lf_prof_"))
    try:
        src, dst = tmp / "images", tmp / "augmented"
        bench._e2e_make_dataset(src, dev, 16, bench._e2e_layout(int(os.environ.get('LF_E2E_IMAGES', '25000'))))
        os.chdir(tmp)
        bal = DatasetBalancer(source_dir=str(src), target_dir=str(dst), seed=42, workers=16)
        bal.analyze_distribution()
        bal.calculate_plan()
        pr = cProfile.Profile()
        pr.enable()
        bal.execute_balancing()
        pr.disable()
        print({k: round(v, 2) for k, v in bal.timings.items()})
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(30)
        st.sort_stats("tottime").print_stats(45)
    finally:
        os.chdir("/")
        shutil.rmtree(tmp, ignore_errors=True)
