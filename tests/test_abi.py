"""The C-ABI library loads and exports every symbol include/leafhip.h declares (no GPU)."""
import re
from pathlib import Path

from leaffliction_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "leafhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in leafhip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) == set(syms)


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.lf_version() == 100
    assert isinstance(lib.lf_last_error(), bytes)


def test_invalid_arguments_are_rejected_without_a_gpu():
    """Argument validation happens on the host before any launch."""
    lib = _lib.load()
    assert lib.lf_hist_u8(None, None, 1, 4, 4, None) == -1
    assert b"null" in lib.lf_last_error()
    assert lib.lf_flip_u8(None, None, None, 0, 4, 4, None) == -1
