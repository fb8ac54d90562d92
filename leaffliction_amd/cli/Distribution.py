"""`Distribution` entrypoint (contract of srcs/cli/Distribution.py:122-196):
`Distribution [ROOT] [--plants P1 P2 ...] [--no-plots]` counts root/PLANT/CLASS/*.jpg, merges the
result into artifacts/plots/distribution.csv and (unless --no-plots) draws one bar and one pie
chart per plant.  Like the reference it logs and RETURNS (exit code 0) on a missing root, an
unknown plant or an empty dataset.  Pure host work."""
from __future__ import annotations

import argparse
import logging
from pathlib import Path
from typing import Optional, Sequence

from ..utils.common import setup_logging
from ..utils.distribution import IMG_EXTS, count_images, merge_csv, plot_per_plant


def parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="Analyze dataset distribution (root/PLANT/CLASS/*.jpg).")
    ap.add_argument("root", nargs="?", default=None, help="Dataset root (default ./images or CWD)")
    ap.add_argument("--plants", nargs="+", default=None, help="Subset of plant names to include")
    ap.add_argument("--no-plots", action="store_true", help="Skip plot generation")
    return ap.parse_args(argv)


def resolve_root(arg_root: Optional[str]) -> Path:
    if arg_root:
        return Path(arg_root)
    default = Path("images")
    return default if default.exists() else Path.cwd()


def main(argv: Optional[Sequence[str]] = None) -> None:
    args = parse_args(argv)
    setup_logging()
    root = resolve_root(args.root)
    if not root.exists():
        logging.error("Root directory does not exist: %s", root)
        return
    available = {p.name for p in root.iterdir() if p.is_dir()}
    wanted = None
    if args.plants:
        wanted = set(args.plants)
        unknown = sorted(wanted - available)
        if unknown:
            for name in unknown:
                logging.warning("Plant directory not found: %s", name)
            logging.error("Aborting due to unknown plant(s). Available: %s", ", ".join(sorted(available)))
            return
    rows = count_images(root, wanted)
    if not rows:
        logging.warning("No images found (supported extensions: %s)", ", ".join(sorted(IMG_EXTS)))
        return
    out_dir = Path("artifacts/plots")
    merge_csv(rows, out_dir / "distribution.csv")
    logging.info("CSV written/updated: %s", (out_dir / "distribution.csv").resolve())
    if not args.no_plots and plot_per_plant(rows, out_dir):
        logging.info("Plots written to: %s", out_dir.resolve())
    logging.info("Total images counted: %d", sum(n for _, _, n in rows))


if __name__ == "__main__":
    main()
