import os
import sys

import pytest

# PyTorch-ROCm ships its own libamdhip64; libdockauv.so links the system one by SONAME.  Whichever is loaded first
# serves both, and a process that ends up with two HIP runtimes loses the device in the second.  bench.py imports
# torch first; do the same for every test process so that tests mixing torch tensors and the C ABI see one runtime.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
