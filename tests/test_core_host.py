"""snappy_amd/csrc/sha512_core.h (the block + padding logic every HIP kernel
shares) compiled as host C++ and checked against the oracle.  CPU only; the
device build of the same header is checked by the -m gpu parity tests."""
import ctypes
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def core(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("core") / "libcorehost.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "core_host_harness.cpp")])
    L = ctypes.CDLL(so)
    L.core_sha512.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p]
    L.core_sha512.restype = None

    def f(data, split=0):
        out = ctypes.create_string_buffer(64)
        L.core_sha512(data, len(data), split, out)
        return out.raw
    return f


def test_every_tail_length(core, oracle):
    rnd = os.urandom(1024)
    for n in range(0, 520):
        assert core(rnd[:n]) == oracle.sha512(rnd[:n]), n


def test_segment_carry(core, oracle):
    rnd = os.urandom(5000)
    for n in (256, 257, 1000, 4096, 4999, 5000):
        for split in (128, 256, 1024):
            if split < n:
                assert core(rnd[:n], split) == oracle.sha512(rnd[:n]), (n, split)
