"""Host edge of the device path: JPEG files <-> uint8 arrays.

Decode and encode stay on host threads (Pillow / libjpeg-turbo), which is where the reference does
them (srcs/utils/image_utils.py:19-59); everything between runs on the GPU through
`leaffliction_amd.ops`.  The names and error behaviour callers rely on are the reference's
(`ImageLoader.*`, `ImageTransforms.*`): only `.jpg` files are images (image_utils.py:13), missing
paths are `FileNotFoundError`, other suffixes `ValueError`, codec failures `RuntimeError`, JPEG
quality 95.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple, Union

import numpy as np
from PIL import Image

PathLike = Union[str, Path]
SUPPORTED_EXTENSIONS = {".jpg"}


def _checked(path: PathLike, *, what: str, need_file: bool = False) -> Path:
    p = Path(path)
    if not p.exists():
        raise FileNotFoundError(f"{what} not found: {p}")
    if need_file and not p.is_file():
        raise ValueError(f"Path is not a file: {p}")
    if p.suffix.lower() not in SUPPORTED_EXTENSIONS:
        raise ValueError(f"Unsupported image format: {p.suffix}")
    return p


def _open_rgb(path: Path, ensure_rgb: bool) -> Image.Image:
    try:
        img = Image.open(path)
        return img.convert("RGB") if ensure_rgb and img.mode != "RGB" else img
    except Exception as exc:  # noqa: BLE001 - the reference folds every codec error into one type
        raise RuntimeError(f"Error loading image {path}: {exc}")


def _write(img: Image.Image, path: PathLike, quality: int) -> None:
    dst = Path(path)
    dst.parent.mkdir(parents=True, exist_ok=True)
    try:
        img.save(dst, quality=quality)   # format from the suffix, default 4:2:0 subsampling
    except Exception as exc:  # noqa: BLE001
        raise RuntimeError(f"Error saving image {dst}: {exc}")


def _to_u8(a: np.ndarray) -> np.ndarray:
    if a.dtype == np.uint8:
        return a
    return (a * 255).astype(np.uint8) if a.max() <= 1.0 else a.astype(np.uint8)


class ImageLoader:
    """Static namespace with the reference's method names."""

    @staticmethod
    def load_pil_image(image_path: PathLike, ensure_rgb: bool = True) -> Image.Image:
        return _open_rgb(_checked(image_path, what="Image"), ensure_rgb)

    @staticmethod
    def load_as_array(image_path: PathLike, ensure_rgb: bool = True) -> np.ndarray:
        """HxWx3 uint8: what the loader threads hand to the pinned staging buffers."""
        return np.array(_open_rgb(_checked(image_path, what="Image"), ensure_rgb))

    @staticmethod
    def save_pil_image(img: Image.Image, output_path: PathLike, quality: int = 95) -> None:
        _write(img, output_path, quality)

    @staticmethod
    def save_array(arr: np.ndarray, output_path: PathLike, quality: int = 95) -> None:
        _write(Image.fromarray(arr), output_path, quality)

    @staticmethod
    def array_to_pil(array: np.ndarray) -> Image.Image:
        return Image.fromarray(_to_u8(array))

    @staticmethod
    def get_image_files(directory: PathLike) -> List[Path]:
        """Sorted matches of the reference's four globs per extension — top-level files are listed
        twice, once by `*.jpg` and once by `**/*.jpg` (image_utils.py:81-88):
        callers count on that multiplicity."""
        root = Path(directory)
        if not root.exists():
            raise FileNotFoundError(f"Directory not found: {root}")
        if not root.is_dir():
            raise ValueError(f"Path is not a directory: {root}")
        patterns = [f"{prefix}*{ext}" for e in SUPPORTED_EXTENSIONS for prefix in ("", "**/")
                    for ext in (e, e.upper())]
        return sorted(p for pattern in patterns for p in root.glob(pattern))

    @staticmethod
    def validate_image_path(image_path: PathLike) -> Path:
        return _checked(image_path, what="Path", need_file=True)


class ImageTransforms:
    """Device versions of resize_image / normalize_array (image_utils.py:106-130)."""

    @staticmethod
    def resize_image(batch_u8, size: Tuple[int, int]):
        """[N,H,W,3] u8 device tensor -> LANCZOS resize to (size[0], size[1]); square only."""
        from .. import ops
        if size[0] != size[1]:
            raise ValueError("resize_image: the hot path only resizes to square img_size")
        return ops.resize_lanczos_u8(batch_u8, int(size[0]))

    @staticmethod
    def normalize_array(batch_u8):
        """[N,H,W,3] u8 device tensor -> [N,3,H,W] f32 = x/255 (NCHW is the device layout)."""
        from .. import ops
        return ops.pack_hwc_u8_to_nchw_f32(batch_u8)
