import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by gpurun)")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import rust_snappy_amd as R
    c = R.raw.Context(0)
    yield c
    c.close()
