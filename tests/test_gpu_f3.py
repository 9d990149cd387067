"""Row f3 (+ f2 fused) on a real MI355X: the block-parallel DEFLATE kernel and the data.tar.gz producer,
through the C ABI.  Checkers: Python's zlib/gzip/tarfile (format validity and exact content), hashlib and
the oracle's hashes.yaml (the fused pass)."""
import gzip
import hashlib
import io
import os
import tarfile
import zlib

import numpy as np
import pytest

import trees
from conftest import ROOT
from test_f3_host import sample_inputs, _make_tree, f3  # noqa: F401  (the CPU model of the kernel's format)

pytestmark = pytest.mark.gpu


@pytest.mark.kernels_only("the DEFLATE kernel alone: nothing in it is planned")
def test_gpu_deflate_round_trips_and_equals_its_cpu_model(built_lib, f3):  # noqa: F811
    """Every sample: gzip.decompress(GPU output) == input, CRC/ISIZE right -- and the bytes equal the serial
    CPU model of the kernel (same chunking, hash chains, lazy parse, encoder), so the workgroup's pipeline is exact."""
    import ctypes
    from snappy_amd import Context
    with Context(staging_bytes=1 << 20) as c:  # 1 MiB staging: inputs cross several staging pieces
        rng = np.random.default_rng(3)
        samples = dict(sample_inputs())
        samples["3 MiB mixed"] = (b"snappy " * 100000 + rng.integers(0, 256, size=1 << 20, dtype=np.uint8).tobytes() +
                                  bytes(1 << 20))[:3 << 20]
        for name, data in samples.items():
            gz = c.gzip_buffer(data)
            assert gzip.decompress(gz) == data, name
            assert int.from_bytes(gz[-8:-4], "little") == zlib.crc32(data), name
            n = ctypes.c_size_t()
            p = f3.f3_model_gzip2(data, len(data), 1 << 20, ctypes.byref(n))  # same staging piece size as the ctx
            model = ctypes.string_at(p, n.value)
            f3.f3_free(p)
            assert gz == model, name
            st = c.targz_stats()
            assert st["tar_bytes"] == len(data) and st["gz_bytes"] == len(gz)
            pieces = [min(1 << 20, len(data) - o) for o in range(0, len(data), 1 << 20)]
            assert st["chunks"] == sum((p + 65535) // 65536 for p in pieces)


@pytest.mark.kernels_only("the DEFLATE kernel alone: nothing in it is planned")
def test_deflate_depth_is_the_models_depth(built_lib, f3):  # noqa: F811
    """snaphash_config.deflate_depth (ABI 4): the producer's effort -- hash-chain links walked per position.  At 8, 32, the
    default (0 = 96 since round 5: the class of the reference's gzip level 9, clickdeb/deb.go:271) and 128 the GPU's bytes
    equal the serial model's at that depth, gzip reads them back, and a deeper walk never writes more on compressible input."""
    import ctypes
    from snappy_amd import Context
    rng = np.random.default_rng(12)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    text = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=400000) % 2000)[:1600000]
    src = b"".join(open(os.path.join(ROOT, "snappy_amd", "csrc", f), "rb").read() for f in ("hostpass.cpp", "walk.cpp", "planner.cpp"))
    sizes = {}
    for depth in (8, 32, 0, 128):
        with Context(staging_bytes=1 << 20, deflate_depth=depth) as c:
            for name, data in (("text", text), ("sources", src)):
                gz = c.gzip_buffer(data)
                assert gzip.decompress(gz) == data, (name, depth)
                n = ctypes.c_size_t()
                p = f3.f3_model_gzip3(data, len(data), 1 << 20, depth, ctypes.byref(n))
                model = ctypes.string_at(p, n.value)
                f3.f3_free(p)
                assert gz == model, (name, depth)
                sizes[(name, depth or 96)] = len(gz)
    for name in ("text", "sources"):
        assert sizes[(name, 8)] > sizes[(name, 32)] > sizes[(name, 96)] >= sizes[(name, 128)], sizes


def test_tar_create_matches_tarfile_view_of_the_tree(built_lib, tmp_path):
    """snaphash_tar_create == tarCreate(data.tar.gz, dir, exclude <dir>/DEBIAN): the .gz inflates to a tar whose
    members are the tree (names './...', root/root, modes, link targets, contents), DEBIAN left out, as the
    reference's TestSnapDebBuild expects ('./usr/bin/foo' listed, no DEBIAN: clickdeb/deb_test.go)."""
    from snappy_amd import Context
    root = str(tmp_path / "src")
    os.makedirs(root)
    files = _make_tree(root)
    out = str(tmp_path / "data.tar.gz")
    with Context() as c:
        _, digest = c.tar_create(out, root, root + "/DEBIAN")
        st = c.targz_stats()
    raw = open(out, "rb").read()
    assert hashlib.sha512(raw).digest() == digest  # archive-sha512 over the bytes produced (build.go:222)
    assert st["gz_bytes"] == len(raw)
    tf = tarfile.open(fileobj=io.BytesIO(raw), mode="r:gz")
    names = tf.getnames()
    assert "./usr/bin/foo" in names and not any("DEBIAN" in n for n in names) and "./a-fifo" not in names
    assert names == sorted(names, key=lambda s: [p.encode() for p in s.split("/")])  # per-directory byte-wise pre-order
    for m in tf.getmembers():
        assert (m.uid, m.gid, m.uname, m.gname) == (0, 0, "root", "root")
        if m.isreg():
            assert tf.extractfile(m).read() == files[m.name[2:]], m.name
    assert tf.getmember("./usr/bin/link").linkname == "foo" and tf.getmember("./usr/bin/foo").mode & 0o777 == 0o755
    # os.Create over an archive that is already there, longer and shorter than the new one: the library overwrites from
    # offset 0 and sets the length at the END of the pass (targz.inc), the result must be the same bytes as on a fresh path
    with Context() as c:
        for old in (os.urandom(len(raw) + 70001), b"short", raw[:len(raw) // 2] + b"x" * 11):
            with open(out, "wb") as f:
                f.write(old)
            _, digest2 = c.tar_create(out, root, root + "/DEBIAN")
            assert open(out, "rb").read() == raw and digest2 == digest
        assert c.tar_create("/dev/null.gz" if os.path.exists("/dev/null.gz") else out, root, root + "/DEBIAN")[1] == digest
    with pytest.raises(Exception):
        with Context() as c:
            c.tar_create(str(tmp_path / "data.tar.xz"), root)  # "unknown compression extension" for anything but .gz here


def test_fused_build_pass_reads_once_and_matches_writehashes(built_lib, oracle, tmp_path, snaphash_mode):
    """tar + gzip + archive digest + per-file SHA-512 + hashes.yaml from ONE read of every file (rows f2 + f3):
    the yaml equals what snaphash_tree AND the oracle compute afterwards from the tree and the archive that
    was written; files cross staging pieces (1 MiB staging) so chaining values travel between launches.
    In the default configuration the two LONG members (3 MiB and 2 MiB in a 7 MiB tree: their SHA-512 chains would outlast
    the pass on the GPU) are hashed by host threads out of the pinned staging buffer -- piece by piece, they span several
    1 MiB slots -- and everything else by the kernels (round 5: MemberHashers in targz.inc)."""
    from snappy_amd import Context
    rng = np.random.default_rng(11)
    sizes = [int(x) for x in rng.integers(0, 300000, size=60)] + [0, 1, 511, 512, 513, (3 << 20) + 77]
    build, _ = trees.make_synthetic_tree(str(tmp_path), sizes + [1])
    # make some content compressible so that both block kinds occur
    with open(os.path.join(build, "d0000", "text.txt"), "wb") as f:
        f.write(b"all work and no play makes jack a dull boy\n" * 50000)
    os.makedirs(os.path.join(build, "DEBIAN"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("Package: x\n")
    out = str(tmp_path / "data.tar.gz")
    with Context(staging_bytes=1 << 20) as c:
        yaml_fused, digest = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
        st, zs = c.stats(), c.targz_stats()
        # every regular file hashed on the GPU out of the tar stream -- but for the two long ones in the default configuration
        hosted = c.stats_ex()["host_streams"] - 1
        assert c.device_stats(0)["streams"] == len(sizes) + 1 - hosted and (hosted == 0) == (snaphash_mode == "gpu_only")
        ex = c.stats_ex()
        if snaphash_mode == "gpu_only":
            assert ex["host_streams"] == 1 and ex["host_bytes"] == os.path.getsize(out)  # the archive digest: one stream, host core
        else:  # ... and the long members
            # (with 1 MiB slots a slot's share of the pass is small: every member of 30 KiB and more is "long" here)
            assert ex["host_streams"] >= 3 and ex["host_bytes"] >= os.path.getsize(out) + (3 << 20) + 77 + 43 * 50000
        assert st["streams"] == len(sizes) + 2 and zs["stored_chunks"] > 0 and zs["stored_chunks"] < zs["chunks"]
        assert c.tree(build, out) == yaml_fused
        assert c.verify(build, yaml_fused, out) is None
    assert oracle.hashes_yaml(build, out) == yaml_fused
    assert hashlib.sha512(open(out, "rb").read()).digest() == digest
    tf = tarfile.open(out, "r:gz")
    for m in tf.getmembers():
        if m.isreg():
            assert tf.extractfile(m).read() == open(os.path.join(build, m.name[2:]), "rb").read(), m.name


def test_tar_create_error_paths_leave_no_archive_and_a_usable_ctx(built_lib, oracle, tmp_path):
    """tarCreate's error behaviour (clickdeb/deb.go:286-289, 331-339: the first Lstat/Open/Copy error ends the walk
    and is returned) through the pipelined producer: an unreadable file in a later staging slot, an output that
    cannot be written (the writer thread's ENOSPC), a name the ustar header cannot hold, an empty tree -- after each
    failure no partial archive is left behind and the same ctx still produces a correct one."""
    from snappy_amd import Context, _lib
    rng = np.random.default_rng(5)
    build, _ = trees.make_synthetic_tree(str(tmp_path), [int(x) for x in rng.integers(1000, 200000, size=40)])
    os.makedirs(os.path.join(build, "DEBIAN"))
    out = str(tmp_path / "data.tar.gz")
    with Context(staging_bytes=1 << 20) as c:
        good_yaml, _ = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
        assert good_yaml == oracle.hashes_yaml(build, out)
        os.unlink(out)
        # 1. a file that cannot be opened (skipped when running as root, who can read anything)
        victim = sorted(p for p in (os.path.join(dp, f) for dp, _, fs in os.walk(build) for f in fs))[-1]
        os.chmod(victim, 0)
        if os.geteuid() != 0:
            with pytest.raises(_lib.SnaphashError) as ei:
                c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            assert ei.value.code == _lib.EIO and os.path.basename(victim) in str(ei.value) and not os.path.exists(out)
        os.chmod(victim, 0o644)
        # 2. the output cannot take the bytes: the writer thread's error comes back, with errno's text
        full = str(tmp_path / "full.tar.gz")
        os.symlink("/dev/full", full)
        with pytest.raises(_lib.SnaphashError) as ei:
            c.tar_create(full, build, build + "/DEBIAN", with_hashes=True)
        assert ei.value.code == _lib.EIO and "No space left" in str(ei.value)
        with pytest.raises(_lib.SnaphashError) as ei:
            c.tar_create(str(tmp_path / "no-such-dir" / "x.tar.gz"), build)
        assert ei.value.code == _lib.EIO
        # 3. names the ustar fields cannot hold travel in PAX extended headers (also across slot boundaries: 1 MiB staging)
        long_dir = os.path.join(build, "n" * 120)
        os.makedirs(long_dir)
        open(os.path.join(long_dir, "m" * 120), "wb").write(b"x" * 70000)
        os.symlink("t" * 180, os.path.join(long_dir, "lnk"))
        yl, _ = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
        assert yl == oracle.hashes_yaml(build, out)
        tfl = tarfile.open(out, "r:gz")
        ml = tfl.getmember("./" + "n" * 120 + "/" + "m" * 120)
        assert tfl.extractfile(ml).read() == b"x" * 70000 and tfl.getmember("./" + "n" * 120 + "/lnk").linkname == "t" * 180
        tfl.close()
        os.unlink(out)
        import shutil
        shutil.rmtree(long_dir)
        # ... and the ctx is as good as new
        y2, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
        assert y2 == oracle.hashes_yaml(build, out) and hashlib.sha512(open(out, "rb").read()).digest() == dig
        # 4. an empty tree: two zero records, `files: []`
        empty = str(tmp_path / "empty")
        os.makedirs(empty)
        out2 = str(tmp_path / "empty.tar.gz")
        y3, dig3 = c.tar_create(out2, empty, empty + "/DEBIAN", with_hashes=True)
        assert gzip.decompress(open(out2, "rb").read()) == bytes(1024)
        assert y3 == oracle.hashes_yaml(empty, out2) and b"files: []" in y3


def test_tar_create_slot_boundaries_random_trees(built_lib, oracle, tmp_path):
    """Slot-boundary logic of the producer under small, odd staging sizes: headers, file bodies, record padding and
    the two closing zero records all get to straddle a slot somewhere in 24 random trees x 3 staging sizes; every
    archive must inflate to the tar stream the host model lays out, and every fused hashes.yaml must be the oracle's."""
    import ctypes
    import shutil
    from snappy_amd import Context
    rng = np.random.default_rng(21)
    so = str(tmp_path / "libf3host.so")
    import subprocess
    from conftest import ROOT
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "f3_host_harness.cpp")])
    L = ctypes.CDLL(so)
    L.f3_tar_stream.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.f3_free.argtypes = [ctypes.c_void_p]
    ctxs = [Context(staging_bytes=s) for s in (1 << 16, (1 << 16) + (1 << 14), 3 << 16)]
    try:
        for it in range(24):
            build = str(tmp_path / ("t%d" % it))
            os.makedirs(os.path.join(build, "DEBIAN"))
            nfiles = int(rng.integers(0, 14))
            for k in range(nfiles):
                d = os.path.join(build, "d%d" % int(rng.integers(0, 3)))
                os.makedirs(d, exist_ok=True)
                # sizes cluster around the record and slot sizes
                size = int(rng.choice([0, 1, 511, 512, 513, 16383, 16384, 65536 - 512, 65536, 65537, int(rng.integers(0, 200000))]))
                data = rng.integers(0, 256, size=size, dtype=np.uint8).tobytes() if k % 2 else (b"abcdefgh" * (size // 8 + 1))[:size]
                with open(os.path.join(d, "f%02d" % k), "wb") as f:
                    f.write(data)
            if it % 3 == 0:
                os.symlink("f00", os.path.join(build, "link"))
            if it % 2 == 1:  # PAX extended headers (names the ustar fields cannot hold) get to straddle slots too
                ld = os.path.join(build, "d1", "L" * int(rng.integers(101, 200)))
                os.makedirs(ld, exist_ok=True)
                with open(os.path.join(ld, "n" * int(rng.integers(101, 250))), "wb") as f:
                    f.write(rng.integers(0, 256, size=int(rng.integers(0, 70000)), dtype=np.uint8).tobytes())
                os.symlink("T" * int(rng.integers(101, 400)), os.path.join(ld, "lnk"))
            p, n = ctypes.c_void_p(), ctypes.c_size_t()
            assert L.f3_tar_stream(build.encode(), (build + "/DEBIAN").encode(), ctypes.byref(p), ctypes.byref(n)) == 0
            want_tar = ctypes.string_at(p.value, n.value)
            L.f3_free(p)
            for c in ctxs:
                out = str(tmp_path / "o.tar.gz")
                y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
                raw = open(out, "rb").read()
                assert gzip.decompress(raw) == want_tar, (it, nfiles)
                assert hashlib.sha512(raw).digest() == dig
                assert y == oracle.hashes_yaml(build, out), (it, nfiles)
            shutil.rmtree(build)
    finally:
        for c in ctxs:
            c.close()


def test_first_slot_goes_to_the_compressor_in_parts(built_lib, oracle, tmp_path):
    """A pass's first slot (default staging: 256 MiB) is read, copied and compressed 64 MiB at a time (targz.inc, in_parts):
    the first chunk kernel runs while the rest of the slot is still on its way.  Members straddle both part boundaries;
    the bytes must equal the serial CPU model's over the same tar stream, inflate to it, and not depend on what the
    staging buffer held before (a different tree goes through the same ctx in between)."""
    import ctypes
    import subprocess
    from snappy_amd import Context
    rng = np.random.default_rng(77)
    so = str(tmp_path / "libf3host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "f3_host_harness.cpp")])
    L = ctypes.CDLL(so)
    L.f3_tar_stream.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.f3_model_gzip2.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.f3_model_gzip2.restype = ctypes.c_void_p
    L.f3_free.argtypes = [ctypes.c_void_p]
    MiB = 1 << 20
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(500)]
    text = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=300000) % 500)

    def content(kind, size):
        if kind == "random":
            return rng.integers(0, 256, size=size, dtype=np.uint8).tobytes()
        off = int(rng.integers(0, len(text) - 1))
        return ((text[off:] + text) * (size // len(text) + 2))[:size]

    def make(root, spec):
        os.makedirs(os.path.join(root, "DEBIAN"))
        open(os.path.join(root, "DEBIAN", "control"), "w").write("Package: parts\n")
        for name, kind, size in spec:
            with open(os.path.join(root, name), "wb") as f:
                f.write(content(kind, size))

    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    make(a, [("f0", "text", 63 * MiB - 700), ("f1", "random", 5 * MiB + 13), ("f2", "random", 60 * MiB + 511), ("f3", "text", 10 * MiB), ("f4", "text", 0), ("f5", "random", 1)])
    make(b, [("g0", "random", 90 * MiB), ("g1", "text", 20 * MiB + 5)])
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    assert L.f3_tar_stream(a.encode(), (a + "/DEBIAN").encode(), ctypes.byref(p), ctypes.byref(n)) == 0
    want_tar = ctypes.string_at(p.value, n.value)
    L.f3_free(p)
    assert 2 * 64 * MiB < len(want_tar) < 256 * MiB  # three parts, one slot
    out = str(tmp_path / "o.tar.gz")
    with Context() as c:
        y1, d1 = c.tar_create(out, a, a + "/DEBIAN", with_hashes=True)
        raw1 = open(out, "rb").read()
        c.tar_create(str(tmp_path / "other.tar.gz"), b, b + "/DEBIAN", with_hashes=True)
        y3, d3 = c.tar_create(out, a, a + "/DEBIAN", with_hashes=True)
        raw3 = open(out, "rb").read()
    assert raw1 == raw3 and d1 == d3 and y1 == y3
    assert hashlib.sha512(raw1).digest() == d1 and y1 == oracle.hashes_yaml(a, out)
    assert gzip.decompress(raw1) == want_tar
    m = ctypes.c_size_t()
    q = L.f3_model_gzip2(want_tar, len(want_tar), 0, ctypes.byref(m))  # one slot = one piece of the model
    model = ctypes.string_at(q, m.value)
    L.f3_free(q)
    assert raw1 == model


def test_tarcreate_as_the_reference_tests_it(built_lib, tmp_path):
    """clickdeb/deb_test.go:160-205 (TestTarCreate) against the mirror of the Go function, snappy_amd.clickdeb.tarCreate:
    the same tree, the same exclude FUNCTION (a suffix rule, not a prefix), the same assertions on `tar tvf` -- with
    ".gz" where the reference's test says ".xz" (upstream shells out to xz for that)."""
    import re
    import subprocess
    from snappy_amd import clickdeb
    builddir = str(tmp_path / "b")
    os.makedirs(os.path.join(builddir, "etc"), mode=0o700)
    for name, data in (("foo", b"foo"), ("exclude-me", b"me")):
        with open(os.path.join(builddir, name), "wb") as f:
            f.write(data)
        os.chmod(os.path.join(builddir, name), 0o644)
    os.symlink("foo", os.path.join(builddir, "link-to-foo"))
    tarname = str(tmp_path / "data.tar.gz")
    asked = []

    def fn(path):
        asked.append(path)
        return not path.endswith("exclude-me")
    clickdeb.tarCreate(tarname, builddir, fn)
    assert os.stat(tarname).st_size != 0                     # "verify that the file is flushed"
    output = subprocess.run(["tar", "tvf", tarname], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True).stdout.decode()
    assert "exclude-me" not in output                        # "exclusion works"
    assert re.search(r"-rw-r--r--[ ]+root/root[ ]+3[ ]+(.*)./foo", output)             # "expected content for the file"
    assert re.search(r"lrwxrwxrwx[ ]+root/root[ ]+0[ ]+(.*)./link-to-foo -> foo", output)  # "and for the symlink"
    assert re.search(r"drwx------[ ]+root/root[ ]+0[ ]+(.*)./etc", output)               # "and for the dir"
    assert not re.search(r"(.*)\.\n", output)                                            # 'and no "." dir'
    # fn is asked for every supported entry, the root included (deb.go:295 comes before the "." test at :310), in walk order
    assert asked == [builddir, builddir + "/etc", builddir + "/exclude-me", builddir + "/foo", builddir + "/link-to-foo"]
    with pytest.raises(Exception):
        clickdeb.tarCreate(str(tmp_path / "x.tar.zz"), builddir, None)  # "unknown compression extension"


@pytest.mark.kernels_only("the DEFLATE kernel alone: nothing in it is planned")
def test_gpu_deflate_equals_model_on_random_structures(built_lib, f3):  # noqa: F811
    """Forty synthetic inputs made of the things a parse can trip over -- copies from every distance up to beyond the
    window, of every length up to beyond 258, runs, literal bursts, cut at awkward sizes around the chunk and tile
    sizes -- through the GPU compressor: gzip inflates each to its input and the bytes equal the CPU model's."""
    import ctypes
    from snappy_amd import Context
    rng = np.random.default_rng(77)
    with Context(staging_bytes=1 << 18) as c:
        for it in range(40):
            target = int(rng.choice([63, 64, 65, 4095, 4096, 4097, 65535, 65536, 65537, 131072 + 5, int(rng.integers(1, 400000))]))
            buf = bytearray()
            while len(buf) < target:
                kind = int(rng.integers(0, 5))
                if kind == 0 or len(buf) < 8:
                    buf += rng.integers(0, 256, size=int(rng.integers(1, 300)), dtype=np.uint8).tobytes()
                elif kind == 1:
                    buf += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 1200))
                elif kind == 2:  # a copy from anywhere behind, any length: overlapping copies included
                    dist = int(rng.integers(1, min(len(buf), 70000) + 1))
                    for _ in range(int(rng.integers(3, 700))):
                        buf.append(buf[-dist])
                elif kind == 3:  # text-like: few symbols
                    buf += bytes(rng.choice(np.frombuffer(b"etaoin shrdlu\n", dtype=np.uint8), size=int(rng.integers(1, 2000))))
                else:            # the same short phrase again and again with one byte changed
                    phrase = bytearray(rng.integers(97, 123, size=int(rng.integers(4, 40)), dtype=np.uint8).tobytes())
                    for _ in range(int(rng.integers(1, 60))):
                        phrase[int(rng.integers(0, len(phrase)))] = int(rng.integers(97, 123))
                        buf += phrase
            data = bytes(buf[:target])
            gz = c.gzip_buffer(data)
            assert gzip.decompress(gz) == data, it
            n = ctypes.c_size_t()
            p = f3.f3_model_gzip2(data, len(data), 1 << 18, ctypes.byref(n))
            model = ctypes.string_at(p, n.value)
            f3.f3_free(p)
            assert gz == model, (it, len(data))


@pytest.mark.gpu
def test_long_members_of_a_package_go_to_host_threads_and_the_buffers_fit_the_job(built_lib, oracle, tmp_path, snaphash_mode):
    """Round 5 (tools/build_small_probe.py): a package as most snaps are -- many small files, a few long ones.  A lone SHA-512
    chain is 44 MB/s on the GPU, and a hashing workgroup beside the compressor costs it a fifth round of workgroups: in the
    default configuration the members are hashed by host threads out of the pinned staging buffer (still one read of every
    file) -- all of them with eight cores or more, the long ones with fewer; SNAPHASH_FLAG_GPU_ONLY keeps every byte on the
    kernels.  hashes.yaml is the oracle's either way, the archive inflates to the tree -- and the staging and output buffers
    are sized for the 9 MiB job, not the engine's 256 MiB staging size (a gigabyte of pinned memory)."""
    from snappy_amd import Context, _lib
    cores = int(_lib.lib().snaphash_usable_cpus())
    rng = np.random.default_rng(31)
    sizes = [int(x) for x in rng.integers(0, 40000, size=150)] + [6 << 20, (3 << 19) + 5, 0, 129]
    build, _ = trees.make_synthetic_tree(str(tmp_path), sizes + [1])
    os.makedirs(os.path.join(build, "DEBIAN"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("Package: x\n")
    out = str(tmp_path / "data.tar.gz")
    with Context() as c:
        for _ in range(2):  # (the second pass reuses the buffers the first one made)
            yaml_fused, digest = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            ex = c.stats_ex()
            if snaphash_mode == "gpu_only":
                assert ex["host_streams"] == 1 and ex["host_bytes"] == os.path.getsize(out)
            elif cores >= 8:  # the archive and every member, the empty one included
                assert ex["host_streams"] == 1 + len(sizes) and ex["host_bytes"] == os.path.getsize(out) + sum(sizes)
            else:  # the archive and the two long members
                assert ex["host_streams"] == 3 and ex["host_bytes"] == os.path.getsize(out) + (6 << 20) + (3 << 19) + 5
            assert oracle.hashes_yaml(build, out) == yaml_fused
            assert hashlib.sha512(open(out, "rb").read()).digest() == digest
        info = c.engine_info(0)
        assert 0 < info["pinned_bytes"] <= 96 << 20, info  # two 16 MiB slots, two output buffers of their size, small tables
    with tarfile.open(out, "r:gz") as tf:
        names = [m.name for m in tf if m.isreg()]
        assert len(names) == len(sizes)
    # one file of 1 MiB: 24 ms of one chain on the GPU, ~1 ms on a core
    one = str(tmp_path / "one")
    os.makedirs(os.path.join(one, "DEBIAN"))
    open(os.path.join(one, "payload.bin"), "wb").write(rng.integers(0, 256, size=1 << 20, dtype=np.uint8).tobytes())
    with Context() as c:
        y, _ = c.tar_create(out, one, one + "/DEBIAN", with_hashes=True)
        assert oracle.hashes_yaml(one, out) == y
        assert c.stats_ex()["host_streams"] == (1 if snaphash_mode == "gpu_only" else 2)
