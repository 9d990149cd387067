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


def test_eight_streams_side_by_side_on_one_core(core, tmp_path_factory):
    """snappy_amd/csrc/hostsha_x8.cpp: a thread of the host pool runs up to eight streams at once, a stream per 64-bit lane
    (AVX-512; one at a time elsewhere).  Memory and files mixed, every length class around the block and the 256 KiB chunk,
    lanes that empty and refill at different times, 1..8 lanes, streams that keep the core to themselves -- every digest
    hashlib's; a file that shrank, grew or is not there fails the call with its errno and its index."""
    import errno
    import hashlib
    import random
    import numpy as np
    so = str(tmp_path_factory.mktemp("core8") / "libcorehost.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "core_host_harness.cpp"), "-pthread"])
    L = ctypes.CDLL(so)
    L.hostsha_many.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64,
                               ctypes.c_uint, ctypes.c_uint64, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    rng = random.Random(31)
    blob = np.random.default_rng(31).integers(0, 256, size=3 << 20, dtype=np.uint8).tobytes()
    d = tmp_path_factory.mktemp("x8files")
    classes = [0, 1, 111, 112, 127, 128, 129, 255, 256, 4095, 65536, (256 << 10) - 1, 256 << 10, (256 << 10) + 1, (256 << 10) + 128, (512 << 10) + 77, 1 << 20]

    serial = [0]

    def run(specs, lanes, alone_from=0):
        n = len(specs)
        ptrs, paths, lens, keep = (ctypes.c_char_p * n)(), (ctypes.c_char_p * n)(), (ctypes.c_uint64 * n)(), []
        for i, (kind, data, claimed) in enumerate(specs):
            lens[i] = len(data) if claimed is None else claimed
            if kind == "mem":
                keep.append(ctypes.create_string_buffer(data, max(len(data), 1)))
                ptrs[i] = ctypes.cast(keep[-1], ctypes.c_char_p)
            else:
                serial[0] += 1
                p = d / ("f%d" % serial[0])
                keep.append(p)
                if kind != "missing":
                    p.write_bytes(data)
                paths[i] = str(p).encode()
        out = ctypes.create_string_buffer(64 * n)
        bad = ctypes.c_int64(-7)
        rc = L.hostsha_many(ptrs, paths, lens, n, lanes, alone_from, out, ctypes.byref(bad))
        return rc, bad.value, [out.raw[64 * i:64 * i + 64] for i in range(n)]

    for trial in range(60):
        n = rng.randrange(1, 24)
        specs = []
        for i in range(n):
            size = rng.choice(classes) if rng.random() < 0.6 else rng.randrange(0, 700000)
            off = rng.randrange(0, len(blob) - size)
            specs.append((rng.choice(["mem", "file"]), blob[off:off + size], None))
        if trial % 3 == 0:
            specs.sort(key=lambda s: -len(s[1]))  # longest first, as the pool hands them out
        rc, bad, digs = run(specs, rng.randrange(1, 9), alone_from=(300000 if trial % 3 == 0 else 0))
        assert rc == 0 and bad == -1
        assert digs == [hashlib.sha512(s[1]).digest() for s in specs], trial
    # the reference's io.Copy reads what is there: a size taken earlier that no longer holds is an error, whichever lane
    base = [("file", blob[:70000], None)] * 5
    for wrong, code in ((("file", blob[:50000], 50001), errno.EIO), (("file", blob[:50000], 49999), errno.EIO),
                        (("file", blob[:300000], 300128), errno.EIO), (("missing", b"", 10), errno.ENOENT)):
        specs = base[:3] + [wrong] + base[3:]
        for lanes in (1, 8):
            rc, bad, _ = run(specs, lanes)
            assert rc == code and bad == 3, (wrong[2], lanes, rc, bad)


def test_eight_streams_under_asan_and_ubsan(tmp_path):
    """tests/asan_hostsha_x8.cpp: the lane scheduler and the AVX-512 block function with every read checked."""
    exe = str(tmp_path / "asan_x8")
    src = [os.path.join(ROOT, "snappy_amd", "csrc", f) for f in ("hostsha.cpp", "hostsha_x8.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                           os.path.join(ROOT, "tests", "asan_hostsha_x8.cpp")] + src + ["-pthread"])
    work = tmp_path / "files"
    work.mkdir()
    r = subprocess.run([exe, str(work)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"asan x8 driver ok" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr.decode()[-3000:])
