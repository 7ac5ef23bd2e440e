import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # build + load the oracle before any test can touch a GPU: afterwards this process must not
    # fork/exec (a child exec from a GPU-initialised process can take the GPU box down)
    import oracle_py
    oracle_py.lib()


@pytest.fixture(scope="session")
def emu_lib():
    """CPU emulation build of the kernel sources (tests/emu); test infrastructure only."""
    from emu.build_emu import build
    return build()


@pytest.fixture(scope="session")
def gpu_lib():
    """the shipped HIP library; the GPU tests must run on it and nothing else"""
    import importlib
    T = importlib.import_module("experimental-tfhe_amd")
    assert os.path.exists(T.DEFAULT_LIB), "libtfhe_amd.so missing: run python experimental-tfhe_amd/build.py"
    return T.DEFAULT_LIB
