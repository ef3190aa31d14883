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
    """The CPU checker.  Rebuilt through `make` where a toolchain is present (a no-op when it is up to date); on a box
    without make / gcc the prebuilt library that travelled with the tree is used as it is, and without either the tests
    that need the checker are skipped with that reason instead of erroring."""
    import shutil
    from oracle import oracle as O
    lib = ROOT / "oracle" / "libtinympc_oracle.so"
    if shutil.which("make") and shutil.which("gcc"):
        O.build(ref=(Path("/root/reference/src/tinympc").is_dir()))
    elif not lib.exists():
        pytest.skip("the CPU checker (oracle/libtinympc_oracle.so) is neither prebuilt nor buildable here: no make / gcc on this box")
    return O


@pytest.fixture(scope="session")
def tinympc():
    import accelerated_tinympc_amd as T
    return T
