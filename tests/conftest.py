import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "kernels_only(reason): a gpu test that runs in the GPU-only configuration alone (it is about the "
                                       "kernels at full size, or names its own flags throughout)")


MODES = ("gpu_only", "planned")


def pytest_generate_tests(metafunc):
    """what snaphash_init(NULL) hands the cgo shim is the PLANNED configuration, so every -m gpu parity
    test runs twice -- once with every byte through the HIP kernels (SNAPHASH_FLAG_GPU_ONLY: the kernels' own parity) and
    once as the library plans the call by default (host threads, AVX-512 lanes and the planner beside the kernels): every
    digest, every hashes.yaml, every archive must come out the same.  Tests marked kernels_only (the full-size config
    2 / 3 / 5 properties, tests that name their flags themselves) run once.  CPU tests are not touched."""
    if metafunc.definition.get_closest_marker("gpu") is None or "snaphash_mode" not in metafunc.fixturenames:
        return
    only = metafunc.definition.get_closest_marker("kernels_only") is not None or os.environ.get("SNAPHASH_TEST_ONE_MODE") == "1"
    metafunc.parametrize("snaphash_mode", MODES[:1] if only else MODES, indirect=True)


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


@pytest.fixture(autouse=True)
def snaphash_mode(built_lib, request):
    """The configuration a Context made without flags gets for this test: "gpu_only" (SNAPHASH_FLAG_GPU_ONLY: every byte
    through the HIP kernels) or "planned" (flags = 0, the library's default).  gpu tests are parametrized over both
    (pytest_generate_tests); anything else gets gpu_only, or what SNAPHASH_TEST_PLANNED=1 asks for."""
    from snappy_amd import _lib
    mode = getattr(request, "param", None) or ("planned" if os.environ.get("SNAPHASH_TEST_PLANNED") == "1" else "gpu_only")
    _lib.Context.DEFAULT_FLAGS = 0 if mode == "planned" else _lib.FLAG_GPU_ONLY
    yield mode
    _lib.Context.DEFAULT_FLAGS = 0


_ctx_cache = {}


@pytest.fixture(scope="session")
def _ctx_cache_owner():
    yield _ctx_cache
    for c in _ctx_cache.values():
        c.close()
    _ctx_cache.clear()


@pytest.fixture(params=["wide", "split", "pair", "auto"])
def ctx(built_lib, request, snaphash_mode, _ctx_cache_owner):
    """A GPU context per kernel variant and configuration, made once per session; only gpu-marked tests may request it.
    (The four-lane QUAD variant is a build option, `make QUAD=1`: a measured negative; its lane
    simulator test stays in the CPU suite.)"""
    from snappy_amd import Context, _lib
    key = (request.param, snaphash_mode)
    if key not in _ctx_cache_owner:
        kern = {"wide": _lib.KERNEL_WIDE, "split": _lib.KERNEL_SPLIT, "pair": _lib.KERNEL_PAIR, "auto": _lib.KERNEL_AUTO}[request.param]
        _ctx_cache_owner[key] = Context(kernel=kern)
    return _ctx_cache_owner[key]
