import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/librefcpu.so), built on demand. Test infrastructure only."""
    from oracle import refcpu
    refcpu.build()
    refcpu.lib()
    return refcpu


@pytest.fixture(scope="session")
def J():
    import jpeg_encoder_amd
    return jpeg_encoder_amd
