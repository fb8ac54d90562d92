"""DatasetBalancer.execute_balancing() at BASELINE configs[2]'s size with the Huffman decoding of the sources on the GPU
(the default) and in the codec workers (LEAFFLICTION_GPU_HUFFMAN=0), same dataset, one run after the other, twice
(development aid; bench.py's augment_end_to_end is the measurement of record).
  python scripts/e2e_balancer_modes.py [generated images, default 100000]"""
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

# LF_MODES="1:1:,0:0:" = (GPU Huffman, GPU noise, extra environment K=V[,K=V]) per run
VARIANTS = [tuple(v.split(":")) for v in os.environ.get("LF_MODES", "1:1:,0:0:").split(";") if v] if os.environ.get("LF_MODES") \
    else [("1", "1", ""), ("0", "0", "")]

if __name__ == "__main__":
    from leaffliction_amd.preprocessing.dataset_balancer import DatasetBalancer
    from leaffliction_amd.utils.system_info import get_available_cores
    generated = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    dev = torch.device("cuda:0")
    cores = min(get_available_cores(), bench.usable_cores())
    tmp = Path(tempfile.mkdtemp(prefix="lf_modes_"))
    try:
        src = tmp / "images"
        bench._e2e_make_dataset(src, dev, cores, bench._e2e_layout(generated))
        os.chdir(tmp)
        for rep in range(int(os.environ.get("LF_MODES_REPS", "1"))):
            for mode, noise, extra in VARIANTS:   # (many runs fill the box's page cache with dirty files)
                os.environ["LEAFFLICTION_GPU_HUFFMAN"] = mode
                os.environ["LEAFFLICTION_GPU_NOISE"] = noise
                for kv in extra.split(",") if extra else []:
                    os.environ[kv.split("=")[0]] = kv.split("=")[1]
                dst = tmp / "augmented"
                bal = DatasetBalancer(source_dir=str(src), target_dir=str(dst), seed=42, workers=cores)
                bal.analyze_distribution()
                bal.calculate_plan()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bal.execute_balancing()
                sec = time.perf_counter() - t0
                pipe = bal.timings.get("decode_kernels_encode", sec) - bal.timings.get("codec_pool_start", 0.0)
                print(json.dumps({"gpu_huffman": mode, "gpu_noise": noise, "extra": extra, "rep": rep, "images_per_sec": round(bal.completed / sec, 1),
                                  "pipeline_images_per_sec": round(bal.completed / pipe, 1), "failed": bal.failed,
                                  "seconds": round(sec, 2),
                                  "stage_seconds": {k: round(v, 2) for k, v in bal.timings.items()}}), flush=True)
                shutil.rmtree(dst, ignore_errors=True)
    finally:
        os.chdir("/")
        shutil.rmtree(tmp, ignore_errors=True)
