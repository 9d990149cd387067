"""ABI 2 of libsnaphash.so on a real MI355X: several engines behind one ctx (LPT shards + digest
gather inside the library), the streaming batch (row f2), the caller-supplied archive digest,
opt-in hybrid scheduling, and a third opinion on a large hashes.yaml (PyYAML + hashlib).
Everything goes through the C ABI; the oracle only checks."""
import hashlib
import os
import stat

import numpy as np
import pytest

from conftest import GOLDEN
import trees

pytestmark = pytest.mark.gpu


def _ragged_sizes(n, seed, top=1 << 20):
    rng = np.random.default_rng(seed)
    s = np.concatenate([rng.integers(0, 4096, size=n // 2), rng.integers(4096, top, size=n - n // 2 - 8),
                        [0, 1, 111, 112, 127, 128, 129, 255]]).astype(np.uint64)
    rng.shuffle(s)
    return s


def test_two_engines_behind_one_ctx_golden_and_oracle(built_lib, oracle, tmp_path):
    """Device list {0, 0}: two engines (own streams, staging buffers, host threads) on the one GPU of
    this box.  The library LPT-shards the file list, hashes both shards concurrently and gathers the
    digests (RCCL needs distinct devices, so the gather is the per-device copy path here; the RCCL
    path has the same interface and is checked against these copies wherever it runs).  The golden
    hashes.yaml and an oracle tree must come out byte for byte."""
    from snappy_amd import Context, _lib
    build, tar = trees.make_simple_tree(str(tmp_path / "g"))
    want = open(os.path.join(GOLDEN, "hashes_simple.yaml"), "rb").read()
    with Context(devices=[0, 0], flags=_lib.FLAG_CHECK_GATHER) as c:
        assert c.tree(build, tar) == want
        ex = c.stats_ex()
        assert ex["n_devices"] == 2 and ex["gather_kind"] == 2
        sizes = _ragged_sizes(400, 21, 1 << 18)
        b2, t2 = trees.make_synthetic_tree(str(tmp_path / "s"), list(sizes) + [70000])
        got = c.tree(b2, t2)
        assert got == oracle.hashes_yaml(b2, t2)
        d0, d1 = c.device_stats(0), c.device_stats(1)
        assert d0["device"] == 0 and d1["device"] == 0
        assert d0["streams"] + d1["streams"] == 401 and min(d0["streams"], d1["streams"]) > 100  # both engines worked
        assert d0["bytes_hashed"] + d1["bytes_hashed"] == c.stats()["bytes_hashed"] == int(sizes.sum()) + 70000
        assert c.verify(b2, got, t2) is None
        # and the batched primitive keeps walk order across shards
        paths = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(b2) for f in fs)
        assert [d.hex() for d in c.sha512_files(paths)] == [oracle.sha512sum(p) for p in paths]


def test_all_visible_devices_and_rccl_gather(built_lib, oracle):
    """Device list {-1}: every visible GPU.  On a 1-GPU box that is one engine (no gather); on a node it
    is the RCCL all-gather with the per-device copies as its parity check (FLAG_CHECK_GATHER)."""
    import torch
    from snappy_amd import Context, _lib
    bufs = [os.urandom(int(n)) for n in _ragged_sizes(300, 22, 1 << 16)]
    with Context(devices=[-1], flags=_lib.FLAG_CHECK_GATHER) as c:
        got = c.sha512_buffers(bufs)
        ex = c.stats_ex()
    assert ex["n_devices"] == torch.cuda.device_count()
    assert ex["gather_kind"] == (0 if ex["n_devices"] == 1 else 1)
    assert ex["n_devices"] == 1 or ex["gather_checked"] == 1
    assert got == [oracle.sha512(b) for b in bufs]


def test_rccl_gather_path_with_one_rank(built_lib, oracle):
    """FLAG_FORCE_GATHER: the digests stay in HBM and go through the library's single-process RCCL path
    (dlopen, ncclCommInitAll, grouped ncclAllGather, D2H of the gathered slab) with one rank, checked against
    the device's own copy -- the same code an 8-GPU ctx runs, on the one GPU this box has."""
    from snappy_amd import Context, _lib
    bufs = [os.urandom(int(n)) for n in _ragged_sizes(200, 26, 1 << 15)]
    with Context(flags=_lib.FLAG_FORCE_GATHER | _lib.FLAG_CHECK_GATHER) as c:
        got = c.sha512_buffers(bufs)
        ex = c.stats_ex()
        assert ex["gather_kind"] == 1 and ex["gather_checked"] == 1, ex
        got2 = c.sha512_buffers(bufs[:17])  # a second call reuses the communicator
    assert got == [oracle.sha512(b) for b in bufs] and got2 == got[:17]


def test_streaming_batch_random_chunkings_vs_oracle(built_lib, oracle):
    """Row f2: the bytes of many files fed chunk by chunk, hash.Hash-style, sequentially per file (the
    tar producer's order) and interleaved across open files; every chunking gives the oracle's digests."""
    from snappy_amd import Context
    rng = np.random.default_rng(31)
    files = [os.urandom(int(n)) for n in _ragged_sizes(200, 23, 1 << 17)] + [b"", os.urandom(3 << 20)]
    want = [oracle.sha512(f) for f in files]
    with Context(staging_bytes=1 << 20) as c:  # small staging: many flushes, streams cross launches
        # (a) sequential producer, io.Copy-sized chunks
        b = c.batch(len(files))
        for i, f in enumerate(files):
            for o in range(0, len(f), 32768):
                b.append(i, f[o:o + 32768])
            b.end(i)
        assert b.finish() == want
        st = c.stats()
        assert st["bytes_hashed"] == sum(len(f) for f in files) and st["launches"] > 5
        # (b) ragged chunk sizes, several files open at once, appends interleaved
        b = c.batch(len(files))
        pos = [0] * len(files)
        open_ = list(range(len(files)))
        while open_:
            i = open_[int(rng.integers(0, min(len(open_), 6)))]
            n = int(rng.choice([0, 1, 7, 127, 128, 129, 1000, 4096, 65536, 200000]))
            b.append(i, files[i][pos[i]:pos[i] + n])
            pos[i] += n
            if pos[i] >= len(files[i]):
                open_.remove(i)
                if rng.integers(0, 2):
                    b.end(i)  # the rest are ended by finish()
        assert b.finish() == want
        # (c) a batch can be abandoned and the ctx used again
        b = c.batch(3)
        b.append(0, b"abc")
        b.abort()
        assert c.sha512_buffers([b"x"])[0].hex().startswith("a4abd4448c49562d")


def test_streaming_batch_guards(built_lib):
    from snappy_amd import Context, SnaphashError, _lib
    with Context() as c:
        b = c.batch(2)
        with pytest.raises(SnaphashError) as e:
            c.sha512_buffers([b"x"])  # one call in flight per ctx
        assert e.value.code == _lib.EINVAL
        b.end(1)
        with pytest.raises(SnaphashError):
            b.append(1, b"more")  # ended
        with pytest.raises(SnaphashError):
            b.append(2, b"x")  # no such stream
        out = b.finish()
        empty = hashlib.sha512(b"").digest()
        assert out == [empty, empty]
    with Context(devices=[0, 0]) as c, pytest.raises(SnaphashError):
        c.batch(1)  # single-device ctx only


def test_tree_with_caller_supplied_archive_digest(built_lib, oracle, tmp_path):
    """snaphash_tree_ex: the Go side hashes data.tar.gz itself (one stream: a host core beats the GPU) and
    hands the digest over; the yaml equals the all-in-one pass, and write != 0 writes DEBIAN/hashes.yaml."""
    from snappy_amd import Context
    sizes = list(_ragged_sizes(120, 24, 1 << 16)) + [123457]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    with Context() as c:
        whole = c.tree(build, tar)
        arch = hashlib.sha512(open(tar, "rb").read()).digest()
        assert c.tree_ex(build, None, arch) == whole == oracle.hashes_yaml(build, tar)
        assert c.stats()["streams"] == 120  # the archive was not hashed by the call
        c.tree_ex(build, None, arch, write=True)
        p = os.path.join(build, "DEBIAN", "hashes.yaml")
        assert open(p, "rb").read() == whole and stat.S_IMODE(os.stat(p).st_mode) == 0o644


def test_verify_needs_an_archive_digest_when_given_an_archive(built_lib, tmp_path):
    from snappy_amd import Context
    build, tar = trees.make_simple_tree(str(tmp_path))
    with Context() as c:
        y = c.tree(build, tar)
        assert c.verify(build, y, tar) is None
        no_arch = b"\n".join(l for l in y.split(b"\n") if not l.startswith(b"archive-sha512"))
        assert c.verify(build, no_arch) is None              # nothing to check the archive against: not asked to
        assert c.verify(build, no_arch, tar) == (6, "archive-sha512")
        short = y.replace(y.split(b"\n")[0], b"archive-sha512: F00F00")
        assert c.verify(build, short, tar) == (6, "archive-sha512")


def test_hybrid_scheduling_is_opt_in_and_bit_exact(built_lib, oracle):
    """host_threads > 0: the few streams whose single-stream GPU time would set the makespan are hashed
    by the library's own host SHA-512 (never the oracle) concurrently with the GPU batch.  Default 0:
    every byte on the GPU.  Digests are the same either way."""
    from snappy_amd import Context, synthetic
    sizes = np.minimum(synthetic.zipf_sizes(3000), np.uint64(48 << 20))  # head 48 MiB: > 1 s alone on the GPU
    bufs = [synthetic.file_bytes(int(n), i) for i, n in enumerate(sizes)]
    total = sum(len(b) for b in bufs)
    with Context() as c:
        import time
        t0 = time.perf_counter()
        gpu_only = c.sha512_buffers(bufs)
        t_gpu = time.perf_counter() - t0
        ex = c.stats_ex()
        assert ex["host_bytes"] == 0 and ex["gpu_bytes"] == total
    with Context(host_threads=8) as c:
        t0 = time.perf_counter()
        hybrid = c.sha512_buffers(bufs)
        t_hyb = time.perf_counter() - t0
        ex = c.stats_ex()
        assert ex["host_streams"] >= 1 and ex["host_bytes"] + ex["gpu_bytes"] == total
        assert ex["host_bytes"] >= 48 << 20  # at least the head went to a host thread
    assert hybrid == gpu_only
    head = int(np.argmax(sizes))
    for i in [head, 0, 1, 2999]:
        assert hybrid[i] == oracle.sha512(bufs[i])
    assert t_hyb < t_gpu  # the point of it: the head no longer sets the makespan


def test_large_tree_third_opinion_pyyaml_hashlib(built_lib, tmp_path):
    """A 1 200-file tree through snaphash_tree, then checked WITHOUT hostpass.cpp's parser and WITHOUT the
    oracle: PyYAML reads the document, os.walk + os.lstat + hashlib recompute every record."""
    import yaml
    from snappy_amd import Context
    sizes = list(_ragged_sizes(1200, 25, 1 << 15)) + [4097]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    os.symlink("f000003.bin", os.path.join(build, "d0000", "link"))
    os.makedirs(os.path.join(build, "DEBIAN"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("x")
    os.chmod(os.path.join(build, "d0000", "f000001.bin"), 0o755)
    with Context() as c:
        doc = yaml.safe_load(c.tree(build, tar))
    assert doc["archive-sha512"] == hashlib.sha512(open(tar, "rb").read()).hexdigest()
    want = []

    def walk(d, rel):
        for name in sorted(os.listdir(d), key=os.fsencode):  # filepath.Walk: byte-wise per directory
            p, r = os.path.join(d, name), (rel + "/" + name if rel else name)
            if ("/" + r).startswith("/DEBIAN"):
                if os.path.isdir(p) and not os.path.islink(p):
                    walk(p, r)
                continue
            st = os.lstat(p)
            kind = "d" if stat.S_ISDIR(st.st_mode) else ("l" if stat.S_ISLNK(st.st_mode) else "f")
            mode = kind + "".join(ch if st.st_mode & (1 << (8 - k)) else "-" for k, ch in enumerate("rwxrwxrwx"))
            rec = {"name": r, "mode": mode}
            if kind == "f":
                rec["size"] = st.st_size
                rec["sha512"] = hashlib.sha512(open(p, "rb").read()).hexdigest()
            want.append(rec)
            if kind == "d":
                walk(p, r)
    walk(build, "")
    assert len(doc["files"]) == len(want) == 1200 + 12 + 1
    assert doc["files"] == want


def test_distinct_contexts_on_concurrent_threads(built_lib, oracle, tmp_path):
    """The threading contract of include/snaphash.h: a ctx serves one call at a time, DISTINCT ctxs
    may be used concurrently.  Four threads, each with its own ctx, run different entry points at once (tree over
    files, host buffers, the fused tar.gz producer, gzip of a buffer); every result is checked against the oracle /
    hashlib / gzip.  ctypes releases the GIL for the duration of each call, so the calls really overlap."""
    import gzip
    import hashlib
    import threading
    from snappy_amd import Context
    rng = np.random.default_rng(9)
    build, tar = trees.make_synthetic_tree(str(tmp_path), [int(x) for x in rng.integers(1, 400000, size=80)])
    os.makedirs(os.path.join(build, "DEBIAN"), exist_ok=True)
    want_yaml = oracle.hashes_yaml(build, tar)
    bufs = [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in rng.integers(0, 300000, size=120)]
    want_digs = [hashlib.sha512(b).digest() for b in bufs]
    text = (b"concurrency is not parallelism " * 40000)[: 1 << 20]
    errors = []

    def guard(fn):
        def run():
            try:
                for _ in range(3):
                    fn()
            except BaseException as e:  # noqa: BLE001 -- report from the main thread
                errors.append(repr(e))
        return run

    def t_tree():
        with Context(staging_bytes=1 << 20) as c:
            assert c.tree(build, tar) == want_yaml

    def t_buffers():
        with Context(staging_bytes=1 << 20) as c:
            assert list(c.sha512_buffers(bufs)) == want_digs

    def t_build():
        out = str(tmp_path / ("o%d.tar.gz" % threading.get_ident()))
        with Context(staging_bytes=1 << 20) as c:
            y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            assert hashlib.sha512(open(out, "rb").read()).digest() == dig and y == oracle.hashes_yaml(build, out)

    def t_gzip():
        with Context(staging_bytes=1 << 18) as c:
            assert gzip.decompress(c.gzip_buffer(text)) == text

    th = [threading.Thread(target=guard(f)) for f in (t_tree, t_buffers, t_build, t_gzip)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in th)


def test_contexts_release_their_device_memory(built_lib, tmp_path):
    """Every allocation a ctx makes lazily (staging slots, deflate scratch incl. the 1 GiB token scratch of the default
    staging size, gather buffers, pinned buffers) goes back on snaphash_destroy: ten create / use / destroy cycles
    leave the device's free memory where it was."""
    import torch
    from snappy_amd import Context
    build, tar = trees.make_synthetic_tree(str(tmp_path), [1000, 70000, 3, 0, 5])
    os.makedirs(os.path.join(build, "DEBIAN"), exist_ok=True)
    out = str(tmp_path / "o.tar.gz")
    torch.cuda.synchronize()

    def cycle():
        with Context(devices=[0, 0], flags=0) as c:  # two engines: both allocate
            c.tree(build, tar)
        with Context() as c:                          # default staging: the large deflate scratch
            c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            c.gzip_buffer(b"abc" * 100000)
    cycle()  # warm: the runtime's own pools settle
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(10):
        cycle()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)  # a leak of any per-ctx buffer would be hundreds of MiB per cycle
