import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/sha512_oracle.c via ctypes."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def built_lib():
    """libsnaphash.so, built in-tree if absent (hipcc cross-compiles without a GPU)."""
    from snappy_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "snappy_amd", "csrc")])
    return _lib.lib()


@pytest.fixture(scope="session", params=["wide", "split", "pair", "quad", "auto"])
def ctx(built_lib, request):
    """A GPU context per kernel variant; only gpu-marked tests may request it."""
    from snappy_amd import Context, _lib
    kern = {"wide": _lib.KERNEL_WIDE, "split": _lib.KERNEL_SPLIT, "pair": _lib.KERNEL_PAIR, "quad": _lib.KERNEL_QUAD,
            "auto": _lib.KERNEL_AUTO}[request.param]
    c = Context(kernel=kern)
    yield c
    c.close()
