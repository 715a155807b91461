import os
import sys

import pytest

# The CPU checkers (oracle/) use OpenMP; on a shared GPU box spinning barriers of 256 threads cost far more than the work
# of the small parity cases. Must be set before libgomp is first loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libmvgx_hip.so and the C oracle exist (cross-compiles here; prebuilt on the GPU box)."""
    from openmvg_amd import build as b
    if not os.path.exists(b.LIB):
        b.build_hip()
    from tests import _oracle
    _oracle.ensure_port()
    return b.LIB
