import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("ER_ORACLE_QUIET", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the C-ABI library and the oracle checkers exist (both build without a GPU)."""
    import __graft_entry__ as g
    from elasticreconstruction_amd import _ffi
    from oracle import pyoracle
    if not os.path.exists(_ffi.LIB_PATH) or not os.path.exists(os.path.join(pyoracle.HERE, "_build", "libtsdf_oracle.so")) \
            or not os.path.exists(os.path.join(pyoracle.HERE, "_build", "libicp_oracle.so")):
        g.build()
    yield


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from elasticreconstruction_amd import _ffi
    assert _ffi.lib().er_device_count() > 0, "HIP extension sees no device although torch does"
    return torch.device("cuda:0")
