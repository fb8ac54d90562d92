import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout; a no-op without the plugin)")


def pytest_collection_modifyitems(config, items):
    """Every test gets a hard limit (pytest-timeout, thread method: it also fires when the main thread is blocked
    inside a C call): a lost rank or codec worker must fail the test, not stall the run for the process group's
    30-minute default."""
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(420, method="thread"))


@pytest.fixture(scope="session")
def golden():
    arrays = np.load(GOLDEN / "augment_golden.npz")
    meta = json.loads((GOLDEN / "augment_golden.json").read_text())
    return arrays, meta


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda is unavailable (no CPU fallback exists)")
    from leaffliction_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def leaf_like(h, w, seed):
    """SURVEY §8d set L: gray background, green disc, brown spots, N(0,8) noise."""
    rng = np.random.RandomState(seed)
    img = np.clip(rng.normal(150, 8, (h, w, 1)).repeat(3, axis=2), 0, 255)
    yy, xx = np.mgrid[0:h, 0:w]
    s = min(h, w) / 224.0
    cy, cx = rng.randint(int(80 * s), int(143 * s) + 1, 2)
    r = rng.randint(int(50 * s), int(89 * s) + 1)
    img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = (60, 140, 50)
    for _ in range(rng.randint(0, 6)):
        by, bx = rng.randint(0, h), rng.randint(0, w)
        br = rng.randint(max(1, int(3 * s)), max(2, int(10 * s)) + 1)
        img[(yy - by) ** 2 + (xx - bx) ** 2 <= br * br] = (120, 70, 30)
    img = img + rng.normal(0, 8, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)
