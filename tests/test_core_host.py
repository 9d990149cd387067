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


def test_round_constants_from_first_principles():
    """K512 / IV512 in sha512_core.h against cube/square roots of the primes (FIPS 180-4 sec. 4.2.3, 5.3.5)."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_sha512_constants import constants
    K, IV = constants()
    text = open(os.path.join(ROOT, "snappy_amd", "csrc", "sha512_core.h")).read()
    body = text[text.index("#define SNAPHASH_K512_LIST"):text.index("static constexpr uint64_t K512")]
    got = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ULL", body)]
    assert got == K
    iv = text[text.index("IV512[8] = {"):]
    assert [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ULL", iv[:iv.index("};")])] == IV


def test_host_sha512_of_the_hybrid_scheduler(core, tmp_path_factory):
    """snappy_amd/csrc/hostsha.cpp (the library's own host SHA-512, used only when host_threads > 0):
    every tail length, resumed mid-stream from a chaining value, and the file reader's size check."""
    import hashlib
    so = str(tmp_path_factory.mktemp("core2") / "libcorehost2.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "core_host_harness.cpp"), "-pthread"])
    L = ctypes.CDLL(so)
    L.hostsha_buffer.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p]
    L.hostsha_file.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]
    rnd = os.urandom(3000)
    out = ctypes.create_string_buffer(64)
    for n in list(range(0, 300)) + [1000, 2047, 2048, 2049, 3000]:
        for split in (0, 128, 256, 1024):
            L.hostsha_buffer(rnd[:n], n, split, out)
            assert out.raw == hashlib.sha512(rnd[:n]).digest(), (n, split)
    # every spelling of the block function this CPU can run agrees with the portable one, block count by block count
    L.hostsha_blocks_variant.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    data = os.urandom(128 * 9)
    for nblocks in range(0, 10):
        got = []
        for v in range(L.hostsha_variants()):
            H = (ctypes.c_uint64 * 8)()
            L.hostsha_blocks_variant(v, data, nblocks, H)
            got.append(list(H))
        assert all(g == got[0] for g in got), nblocks
    p = tmp_path_factory.mktemp("f") / "blob"
    blob = os.urandom((1 << 20) + 77)
    p.write_bytes(blob)
    assert L.hostsha_file(str(p).encode(), len(blob), out) == 0 and out.raw == hashlib.sha512(blob).digest()
    import errno
    assert L.hostsha_file(str(p).encode(), len(blob) - 1, out) == errno.EIO  # grew since its size was taken
    assert L.hostsha_file(str(p).encode(), len(blob) + 1, out) == errno.EIO  # shrank
    assert L.hostsha_file(b"/nonexistent/x", 0, out) == errno.ENOENT
    # a stream of 32 MiB or more is read by a second thread, two buffers ahead of the hasher: same digest, same size checks
    big = tmp_path_factory.mktemp("g") / "big"
    data = os.urandom(1 << 20) * 40 + b"tail"
    big.write_bytes(data)
    for _ in range(3):
        assert L.hostsha_file(str(big).encode(), len(data), out) == 0 and out.raw == hashlib.sha512(data).digest()
    assert L.hostsha_file(str(big).encode(), len(data) - 1, out) == errno.EIO
    assert L.hostsha_file(str(big).encode(), len(data) + (1 << 20), out) == errno.EIO
