"""Image file I/O at the edge of the device path (mirror of srcs/utils/image_utils.py).

JPEG decode/encode stays on the host (Pillow/libjpeg-turbo) exactly as in the reference
(image_utils.py:19-59); everything between decode and encode runs on the GPU through
`leaffliction_amd.ops`.  Only `.jpg` is accepted, like the reference (image_utils.py:13).
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple, Union

import numpy as np
from PIL import Image

SUPPORTED_EXTENSIONS = {".jpg"}


class ImageLoader:
    @staticmethod
    def load_pil_image(image_path: Union[str, Path], ensure_rgb: bool = True) -> Image.Image:
        image_path = Path(image_path)
        if not image_path.exists():
            raise FileNotFoundError(f"Image not found: {image_path}")
        if image_path.suffix.lower() not in SUPPORTED_EXTENSIONS:
            raise ValueError(f"Unsupported image format: {image_path.suffix}")
        try:
            img = Image.open(image_path)
            if ensure_rgb and img.mode != "RGB":
                img = img.convert("RGB")
            return img
        except Exception as e:
            raise RuntimeError(f"Error loading image {image_path}: {e}")

    @staticmethod
    def load_as_array(image_path: Union[str, Path], ensure_rgb: bool = True) -> np.ndarray:
        return np.array(ImageLoader.load_pil_image(image_path, ensure_rgb))

    @staticmethod
    def save_pil_image(img: Image.Image, output_path: Union[str, Path], quality: int = 95) -> None:
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        try:
            img.save(output_path, quality=quality)
        except Exception as e:
            raise RuntimeError(f"Error saving image {output_path}: {e}")

    @staticmethod
    def save_array(arr: np.ndarray, output_path: Union[str, Path], quality: int = 95) -> None:
        ImageLoader.save_pil_image(Image.fromarray(arr), output_path, quality)

    @staticmethod
    def array_to_pil(array: np.ndarray) -> Image.Image:
        if array.dtype != np.uint8:
            if array.max() <= 1.0:
                array = (array * 255).astype(np.uint8)
            else:
                array = array.astype(np.uint8)
        return Image.fromarray(array)

    @staticmethod
    def get_image_files(directory: Union[str, Path]) -> List[Path]:
        directory = Path(directory)
        if not directory.exists():
            raise FileNotFoundError(f"Directory not found: {directory}")
        if not directory.is_dir():
            raise ValueError(f"Path is not a directory: {directory}")
        image_files: List[Path] = []
        for ext in SUPPORTED_EXTENSIONS:  # same four globs as the reference (duplicates included)
            image_files.extend(directory.glob(f"*{ext}"))
            image_files.extend(directory.glob(f"*{ext.upper()}"))
            image_files.extend(directory.glob(f"**/*{ext}"))
            image_files.extend(directory.glob(f"**/*{ext.upper()}"))
        return sorted(image_files)

    @staticmethod
    def validate_image_path(image_path: Union[str, Path]) -> Path:
        image_path = Path(image_path)
        if not image_path.exists():
            raise FileNotFoundError(f"Path not found: {image_path}")
        if not image_path.is_file():
            raise ValueError(f"Path is not a file: {image_path}")
        if image_path.suffix.lower() not in SUPPORTED_EXTENSIONS:
            raise ValueError(f"Unsupported image format: {image_path.suffix}")
        return image_path


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
