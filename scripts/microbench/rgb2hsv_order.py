"""Is the RGB->HSV rate of bench.py's augment table a property of the kernel or of what ran before it?
(round 2: 0.771 of 8 TB/s in r02_a, 0.666-0.692 from r02_c on, with no change to the kernel in between: the only
thing that changed in front of it was the Gaussian blur moving to the i8 matrix cores.)  Runs the same measurement
alone, after each candidate predecessor, and after a pause."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from leaffliction_amd import _lib
    _lib.load()
    rows = []
    for label, only in [("alone", ["rgb2hsv"]),
                        ("after mask_composite", ["mask_composite", "rgb2hsv"]),
                        ("after gauss_blur_5x5", ["gauss_blur_5x5", "rgb2hsv"]),
                        ("after gauss_blur_15x15", ["gauss_blur_15x15", "rgb2hsv"]),
                        ("after both blurs", ["gauss_blur_5x5", "gauss_blur_15x15", "rgb2hsv"]),
                        ("table order up to rgb2hsv", ["flip", "rotate", "skew", "shear", "crop", "distortion", "pack", "hist",
                                                       "mask_composite", "gauss_blur_5x5", "gauss_blur_15x15", "rgb2hsv"]),
                        ("alone again", ["rgb2hsv"])]:
        r = bench.augment_throughput(dev, only=only)
        rows.append((label, r["rgb2hsv"]["frac_hbm_8TBs"], r["rgb2hsv"]["GB_s"]))
        time.sleep(0.5)
    for label, frac, gbs in rows:
        print(f"rgb2hsv {label:28s} {gbs:8.1f} GB/s  {frac:.3f} of 8 TB/s")
    # iterations: 5 (the table's) against 50 back to back
    for it in (5, 50):
        r = bench.augment_throughput(dev, iters=it, only=["rgb2hsv"])
        print(f"rgb2hsv alone, {it:2d} iterations       {r['rgb2hsv']['GB_s']:8.1f} GB/s  {r['rgb2hsv']['frac_hbm_8TBs']:.3f}")


if __name__ == "__main__":
    main()
