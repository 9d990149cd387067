"""Row f4: helpers.FilesAreEqual / DirUpdated (reference helpers/cmp.go).  The oracle's restatement
is pinned by the reference's own test cases (helpers/cmp_test.go:30-140, recreated here); the GPU
path is then checked against the oracle and the same cases."""
import os

import numpy as np
import pytest


def _w(path, data=b""):
    with open(path, "wb") as f:
        f.write(data)


# ---- the reference's cases, as (builder, expected) -- shared by the oracle and the GPU tests ------

def case_cmp_growing(tmp):      # TestCmp: a file equals itself at every length 0, 16, ..., 17 600 (crosses 16 KiB)
    foo = os.path.join(tmp, "foo")
    out = []
    for i in (0, 1, 2, 1023, 1024, 1025, 1099):
        _w(foo, b"*" * (16 * i))
        out.append(((foo, foo), True))
        yield out[-1]


def cases_static(tmp):
    foo, bar = os.path.join(tmp, "foo"), os.path.join(tmp, "bar")
    _w(foo)
    yield (foo, bar), False            # TestCmpEmptyNeqMissing
    yield (bar, foo), False
    _w(bar, b"x")
    yield (foo, bar), False            # TestCmpEmptyNeqNonEmpty
    yield (bar, foo), False
    for a, b, r in ((b"hello", b"hello", True), (b"hello", b"world", False), (b"hello", b"hell", False)):  # TestCmpStreams
        pa, pb = os.path.join(tmp, "sa"), os.path.join(tmp, "sb")
        _w(pa, a)
        _w(pb, b)
        yield (pa, pb), r


def dir_cases(tmp):
    """-> [(d1, d2, pfx, expected set)] following TestDirUpdated*."""
    out = []
    def mk(name):
        p = os.path.join(tmp, name)
        os.makedirs(p)
        return p
    d1, d2 = mk("e1"), mk("e2")
    out.append((d1, d2, "", set()))                                   # EmptyOK
    d1, d2 = mk("x1"), mk("x2")
    _w(os.path.join(d2, "foo"), b"x")
    out += [(d1, d2, "", set()), (d2, d1, "", set())]                 # ExtraFileIgnored (either side)
    d1, d2 = mk("q1"), mk("q2")
    _w(os.path.join(d1, "foo"), b"x")
    _w(os.path.join(d2, "foo"), b"x")
    out.append((d1, d2, "", set()))                                   # FilesEqual
    os.mkdir(os.path.join(d1, "dir"))
    out.append((d1, d2, "", set()))                                   # DirIgnored
    d1, d2 = mk("a1"), mk("a2")
    _w(os.path.join(d1, "foo"), b"y")
    _w(os.path.join(d2, "foo"), b"x")
    _w(os.path.join(d1, "bar"), b"x")
    _w(os.path.join(d2, "bar"), b"y")
    _w(os.path.join(d2, "baz"), b"x")
    out += [(d1, d2, "", {"bar", "foo"}), (d1, d2, "foo_", {"foo_bar", "foo_foo"})]  # AllDifferentReturned
    return out


def test_oracle_pinned_by_reference_cases(oracle, tmp_path):
    tmp = str(tmp_path)
    for (a, b), want in list(case_cmp_growing(tmp)):
        assert oracle.files_equal(a, b) is want
    for (a, b), want in cases_static(tmp):
        assert oracle.files_equal(a, b) is want, (a, b)
    for d1, d2, pfx, want in dir_cases(tmp):
        assert oracle.dir_updated(d1, d2, pfx) == want


def test_oracle_tail_and_buffer_boundaries(oracle, tmp_path):
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    base = os.urandom(3 * 16384 + 5)
    for n in (16383, 16384, 16385, 32768, 3 * 16384 + 5):
        _w(a, base[:n])
        _w(b, base[:n])
        assert oracle.files_equal(a, b)
        for flip in (0, n // 2, n - 1):
            _w(b, base[:flip] + bytes([base[flip] ^ 1]) + base[flip + 1:n])
            assert not oracle.files_equal(a, b), (n, flip)


@pytest.mark.gpu
def test_gpu_reference_cases(ctx, tmp_path):
    from snappy_amd import FilesAreEqual, DirUpdated
    tmp = str(tmp_path)
    for (a, b), want in list(case_cmp_growing(tmp)):
        assert FilesAreEqual(a, b, ctx) is want
    for (a, b), want in cases_static(tmp):
        assert FilesAreEqual(a, b, ctx) is want, (a, b)
    for d1, d2, pfx, want in dir_cases(tmp):
        assert set(DirUpdated(d1, d2, pfx, ctx)) == want


@pytest.mark.gpu
def test_gpu_batch_vs_oracle(built_lib, oracle, tmp_path):
    """Random pairs: equal, one flipped byte anywhere (first, middle, last, in the masked tail piece),
    different sizes, missing, directory; batched through small staging so pairs span batches."""
    from snappy_amd import Context
    rng = np.random.default_rng(12)
    blob = rng.integers(0, 256, size=1 << 20, dtype=np.uint8).tobytes()
    pairs = []
    for i in range(300):
        n = int(rng.choice([0, 1, 15, 16, 17, 255, 4096, 70000, 262144, 262145, 600000]))
        o = int(rng.integers(0, len(blob) - n + 1))
        data = blob[o:o + n]
        a, b = str(tmp_path / ("a%d" % i)), str(tmp_path / ("b%d" % i))
        _w(a, data)
        kind = int(rng.integers(0, 6))
        if kind == 0 or n == 0:
            _w(b, data)
        elif kind == 1:
            k = int(rng.choice([0, n // 2, n - 1]))
            _w(b, data[:k] + bytes([data[k] ^ 0x80]) + data[k + 1:])
        elif kind == 2:
            _w(b, data + b"!")
        elif kind == 3:
            pass  # b missing
        elif kind == 4:
            os.mkdir(b)
        else:
            _w(b, data)
        pairs.append((a, b))
    want = [oracle.files_equal(a, b) for a, b in pairs]
    assert any(want) and not all(want)
    for staging in (1 << 16, 1 << 20, 0):
        with Context(staging_bytes=staging) as c:
            assert c.files_equal(pairs) == want, staging


@pytest.mark.gpu
@pytest.mark.kernels_only("the HBM-resident comparison never plans")
def test_gpu_ranges_equal_device(ctx):
    import torch
    rng = np.random.default_rng(5)
    lens = np.array([0, 1, 15, 16, 17, 4096, 262144, 262145, 1 << 20, 3000001], dtype=np.uint64)
    pad = (lens + np.uint64(15)) // np.uint64(16) * np.uint64(16)
    off = np.concatenate([[0], np.cumsum(pad[:-1])]).astype(np.uint64)
    total = int(off[-1] + pad[-1])
    host_a = rng.integers(0, 256, size=total, dtype=np.uint8)
    host_b = host_a.copy()
    host_b[int(off[3]) + 15] ^= 1                      # last byte of a 16-byte range
    host_b[int(off[4]) + 17] ^= 1                      # just PAST a 17-byte range: must be ignored
    host_b[int(off[7]) + 262144] ^= 1                  # the single byte of a second chunk
    host_b[int(off[9]) + 1500000] ^= 0x10
    a, b = torch.from_numpy(host_a).cuda(), torch.from_numpy(host_b).cuda()
    out = torch.zeros(len(lens), dtype=torch.uint8, device="cuda")
    ctx.ranges_equal_device(a.data_ptr(), off, b.data_ptr(), off, lens, out.data_ptr())
    ctx.sync()
    assert out.cpu().numpy().tolist() == [1, 1, 1, 0, 1, 1, 1, 0, 1, 0]


@pytest.mark.gpu
def test_small_comparisons_stay_on_the_host_by_default(built_lib, oracle, tmp_path, snaphash_mode):
    """Round 5 (tools/cmp_small_probe.py): what policy.AppArmorDelta hands DirUpdated is a few dozen profile files of a few
    KiB (policy/policy.go:162 -> helpers/cmp.go:81-114).  In the default configuration such a job is read and compared by host
    threads -- no staging buffer is pinned for it -- and with SNAPHASH_FLAG_GPU_ONLY it goes through the compare kernel with
    staging halves sized for the job; the verdicts are the oracle's either way."""
    from snappy_amd import Context
    rng = np.random.default_rng(8)
    pairs = []
    for i in range(40):
        data = rng.integers(0, 256, size=int(rng.integers(1, 9000)), dtype=np.uint8).tobytes()
        a, b = str(tmp_path / ("p%d" % i)), str(tmp_path / ("q%d" % i))
        _w(a, data)
        _w(b, data if i % 5 else data[:-1] + bytes([data[-1] ^ 1]))
        pairs.append((a, b))
    pairs.append((pairs[0][0], str(tmp_path / "missing")))
    want = [oracle.files_equal(a, b) for a, b in pairs]
    assert any(want) and not all(want)
    with Context() as c:
        assert c.files_equal(pairs) == want
        pinned = c.engine_info(0)["pinned_bytes"]
        if snaphash_mode == "gpu_only":
            assert 0 < pinned <= 24 << 20, pinned  # two 8 MiB slots and the verdict buffers, not two of 256 MiB
        else:
            assert pinned <= 8 << 20, pinned  # (what the link probe of a planning ctx left behind at init, nothing more)
