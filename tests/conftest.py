import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build(ref=(Path("/root/reference/src/tinympc").is_dir()))
    return O


@pytest.fixture(scope="session")
def tinympc():
    import accelerated_tinympc_amd as T
    return T
