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


@pytest.fixture(scope="session", autouse=True)
def _gpu_only_unless_asked(built_lib):
    """The suite is about the HIP kernels: a Context made without flags keeps every byte on the GPU
    (SNAPHASH_FLAG_GPU_ONLY).  The library's own default -- a stream that would set the makespan of its batch all by
    itself is hashed on a host thread -- is what the tests that pass flags=0 explicitly cover."""
    from snappy_amd import _lib
    # SNAPHASH_TEST_PLANNED=1: run the suite in the library's default (planned) configuration instead -- every digest and
    # every hashes.yaml must come out the same; only the tests that assert WHERE the bytes were hashed differ
    _lib.Context.DEFAULT_FLAGS = 0 if os.environ.get("SNAPHASH_TEST_PLANNED") == "1" else _lib.FLAG_GPU_ONLY
    yield
    _lib.Context.DEFAULT_FLAGS = 0


@pytest.fixture(scope="session", params=["wide", "split", "pair", "auto"])
def ctx(built_lib, request):
    """A GPU context per kernel variant; only gpu-marked tests may request it.  (The four-lane QUAD variant is a
    build option, `make QUAD=1`: a measured negative; its lane simulator test stays in the CPU suite.)"""
    from snappy_amd import Context, _lib
    kern = {"wide": _lib.KERNEL_WIDE, "split": _lib.KERNEL_SPLIT, "pair": _lib.KERNEL_PAIR, "auto": _lib.KERNEL_AUTO}[request.param]
    c = Context(kernel=kern)
    yield c
    c.close()
