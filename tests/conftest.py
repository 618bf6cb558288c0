import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library iterates images whose potentials fit the GPU's last-level cache by half-sweeps (faster
# there).  The tests are about the fused kernels too, on small images: keep them on unless a test asks
# for the default (tests/test_gpu_small_images.py).  Child processes (CLIs, ranks) inherit this.
os.environ.setdefault("DS_CACHE_RESIDENT", "fused")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
