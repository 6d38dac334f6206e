import os
import sys
from pathlib import Path

import pytest

# torch bundles its own HIP runtime; load it before liborbgpu.so so that one libamdhip64 serves the process
# (tests use torch only for device buffers and torch.distributed)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orb_oracle as oo
    oo.build()
    oo.lib()
    return oo
