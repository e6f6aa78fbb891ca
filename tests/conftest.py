import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(ROOT / "tests" / "golden" / "simple_ocp_croco_results.npz")


@pytest.fixture(scope="session")
def panda():
    from agimus_controller_amd.factory import robot_tables as rt

    return rt.panda_table()


@pytest.fixture(scope="session")
def hip_backend():
    """The product library on a real device; GPU tests fail loudly without it."""
    from agimus_controller_amd import backend

    backend.lib()
    assert backend.device_count() > 0, "no HIP device visible"
    return backend
