import os
import sys

import pytest

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
