import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def hostemu_lib():
    import ctypes
    from tests.hostemu import build_hostemu
    lib = ctypes.CDLL(build_hostemu.build())
    lib.emu_model_create.restype = ctypes.c_void_p
    return lib
