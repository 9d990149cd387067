"""ABI 2 of libsnaphash.so on a real MI355X: several engines behind one ctx (LPT shards + digest
gather inside the library), the streaming batch (row f2), the caller-supplied archive digest,
opt-in hybrid scheduling, and a third opinion on a large hashes.yaml (PyYAML + hashlib).
Everything goes through the C ABI; the oracle only checks."""
import hashlib
import os
import stat

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
import trees

pytestmark = pytest.mark.gpu


def _ragged_sizes(n, seed, top=1 << 20):
    rng = np.random.default_rng(seed)
    s = np.concatenate([rng.integers(0, 4096, size=n // 2), rng.integers(4096, top, size=n - n // 2 - 8),
                        [0, 1, 111, 112, 127, 128, 129, 255]]).astype(np.uint64)
    rng.shuffle(s)
    return s


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_two_engines_behind_one_ctx_golden_and_oracle(built_lib, oracle, tmp_path):
    """Device list {0, 0}: two engines (own streams, staging buffers, host threads) on the one GPU of
    this box.  The library LPT-shards the file list, hashes both shards concurrently and gathers the
    digests (RCCL needs distinct devices, so the gather is the per-device copy path here; the RCCL
    path has the same interface and is checked against these copies wherever it runs).  The golden
    hashes.yaml and an oracle tree must come out byte for byte."""
    from snappy_amd import Context, _lib
    build, tar = trees.make_simple_tree(str(tmp_path / "g"))
    want = open(os.path.join(GOLDEN, "hashes_simple.yaml"), "rb").read()
    with Context(devices=[0, 0], flags=_lib.FLAG_CHECK_GATHER | _lib.FLAG_GPU_ONLY) as c:
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


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_all_visible_devices_and_rccl_gather(built_lib, oracle):
    """Device list {-1}: every visible GPU.  On a 1-GPU box that is one engine (no gather); on a node it
    is the RCCL all-gather with the per-device copies as its parity check (FLAG_CHECK_GATHER)."""
    import torch
    from snappy_amd import Context, _lib
    bufs = [os.urandom(int(n)) for n in _ragged_sizes(300, 22, 1 << 16)]
    with Context(devices=[-1], flags=_lib.FLAG_CHECK_GATHER | _lib.FLAG_GPU_ONLY) as c:
        got = c.sha512_buffers(bufs)
        ex = c.stats_ex()
    assert ex["n_devices"] == torch.cuda.device_count()
    assert ex["gather_kind"] == (0 if ex["n_devices"] == 1 else 1)
    assert ex["n_devices"] == 1 or ex["gather_checked"] == 1
    assert got == [oracle.sha512(b) for b in bufs]


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_rccl_gather_path_with_one_rank(built_lib, oracle):
    """FLAG_FORCE_GATHER: the digests stay in HBM and go through the library's single-process RCCL path
    (dlopen, ncclCommInitAll, grouped ncclAllGather, D2H of the gathered slab) with one rank, checked against
    the device's own copy -- the same code an 8-GPU ctx runs, on the one GPU this box has."""
    from snappy_amd import Context, _lib
    bufs = [os.urandom(int(n)) for n in _ragged_sizes(200, 26, 1 << 15)]
    with Context(flags=_lib.FLAG_FORCE_GATHER | _lib.FLAG_CHECK_GATHER | _lib.FLAG_GPU_ONLY) as c:
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
    with Context(devices=[0, 0]) as c:  # the ctx the cgo shim creates ({-1} = all GPUs): the batch runs on its first engine
        b = c.batch(3)
        b.append(0, b"abc")
        b.append(2, b"x" * 1000)
        assert b.finish() == [hashlib.sha512(b"abc").digest(), hashlib.sha512(b"").digest(), hashlib.sha512(b"x" * 1000).digest()]


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


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_hybrid_scheduling_is_bit_exact(built_lib, oracle):
    """host_threads > 0: the few streams whose single-stream GPU time would set the makespan are hashed
    by the library's own host SHA-512 (never the oracle) concurrently with the GPU batch.
    SNAPHASH_FLAG_GPU_ONLY (the suite's default): every byte on the GPU.  Digests are the same either way."""
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
    with Context(host_threads=8, flags=0) as c:
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


@pytest.mark.kernels_only("names the configuration of every Context it makes")
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


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_default_configuration_plans_every_call(built_lib, oracle, tmp_path):
    """snaphash_init(NULL) / flags 0 (ABI 4): every call is planned (planner.cpp).  The package's own archive next to its
    tree (snappy/build.go:222 -- ONE stream, 44 MB/s on the GPU against 1.4 GB/s on a host core) is hashed on a host
    thread while the kernels take the tree; a batch the host alone finishes sooner than any split -- a lone file, a few
    dozen members -- runs on host threads whole (the library's own SHA-512, hostsha.cpp, never the oracle); a large batch
    of similar streams stays on the GPU but for the share the spare cores can take.  hashes.yaml and the digests are the
    oracle's every time."""
    import time
    from snappy_amd import Context, _lib, synthetic
    sizes = [1 << 20] * 48 + [4096, 0, 77]
    build, _ = trees.make_synthetic_tree(str(tmp_path), sizes)
    tar = os.path.join(str(tmp_path), "big.tar.gz")
    with open(tar, "wb") as f:
        f.write(synthetic.file_bytes(96 << 20, 4242))
    want = oracle.hashes_yaml(build, tar)
    with Context(flags=0) as c:
        t0 = time.perf_counter()
        assert c.tree(build, tar) == want
        t_default = time.perf_counter() - t0
        ex = c.stats_ex()
        assert ex["host_streams"] >= 1 and ex["host_bytes"] >= 96 << 20  # the archive for certain
        assert ex["host_bytes"] + ex["gpu_bytes"] == (96 << 20) + sum(sizes[:-1])  # (make_synthetic_tree writes the last size as its own archive stand-in)
        # a few dozen members: the host alone beats the kernels' 24 ms per MiB of the longest member
        paths = [os.path.join(dp, f) for dp, _, fs in os.walk(build) for f in fs]
        t0 = time.perf_counter()
        got = c.sha512_files(paths)
        t_small = time.perf_counter() - t0
        cores = _lib.lib().snaphash_usable_cpus()
        if cores >= 4:  # (one core alone needs 40 ms for these: there the kernels' 24 ms win, and the planner says so)
            assert c.stats_ex()["gpu_bytes"] == 0 and c.stats()["launches"] == 0
        assert got == [oracle.sha512(open(p, "rb").read()) for p in paths]
        assert t_small < (0.0235 if cores >= 4 else 0.05)  # GPU only: 23.8 ms for the 1 MiB members alone (16 cores: ~3 ms)
        # the literal helpers.Sha512sum call: one file, the calling thread, no launch
        t0 = time.perf_counter()
        assert c.sha512_buffers([b"x"]) == [hashlib.sha512(b"x").digest()]
        t_one = time.perf_counter() - t0
        assert c.stats_ex()["host_bytes"] == 1 and c.stats()["launches"] == 0 and t_one < 0.005
        # many similar streams, more than the host could take: the kernels keep most of them
        n = 2048
        blob = np.random.default_rng(3).integers(0, 256, size=(n << 20) + 4096, dtype=np.uint8)
        bufs = [blob[(i << 20) + i % 4096:((i + 1) << 20) + i % 4096] for i in range(n)]
        got = c.sha512_buffers(bufs)
        ex = c.stats_ex()
        cpus = int(_lib.lib().snaphash_usable_cpus())  # what the host may take depends on the cores this job may keep busy
        lane_gain = c.plan_model(False)["host_lane_gain_pct"] / 100.0  # eight streams a thread in AVX-512 lanes: ~3.2 x one stream's 1.4 GB/s
        host_share_bound = min(0.95, cpus * 1.5e9 * lane_gain / (50e9 + cpus * 1.5e9 * lane_gain) + 0.10)
        assert ex["host_bytes"] + ex["gpu_bytes"] == n << 20, (ex, c.plan_model(False), c.calib())
        if ex["gpu_bytes"] == 0:
            # sixteen cores with eight AVX-512 lanes each outrun the link once a core does ~1.8 GB/s of SHA-512 as the model has
            # it (measured rate x what earlier host parts of this ctx did against their plan): met on one box of the pool.  Then
            # the plan must have said so, and the call must not have taken much longer than it said.
            assert ex["planned_gpu_ms"] == 0 and ex["planned_host_ms"] > 0 and ex["host_ms"] < 2.0 * ex["planned_host_ms"], (ex, c.plan_model(False), c.calib())
        else:
            assert ex["host_bytes"] <= host_share_bound * (n << 20), (ex, cpus)
        for i in (0, 1, n // 2, n - 1):
            assert got[i] == hashlib.sha512(bufs[i].tobytes()).digest()
    with Context() as c:  # the suite's default: SNAPHASH_FLAG_GPU_ONLY
        t0 = time.perf_counter()
        assert c.tree(build, tar) == want
        t_gpu_only = time.perf_counter() - t0
        assert c.stats_ex()["host_bytes"] == 0
    assert t_default < t_gpu_only / 3  # 96 MiB alone on the GPU: ~2.2 s


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_default_configuration_through_tree_verify_and_the_producer(built_lib, oracle, tmp_path):
    """What a cgo caller gets from snaphash_init(NULL): the tree, verify and tar_create parity checks once more with
    flags = 0 (the rest of the suite keeps every byte on the GPU).  A tree large enough that the kernels and the
    host threads both have work."""
    import gzip
    import io
    import tarfile
    from snappy_amd import Context
    sizes = list(_ragged_sizes(900, 11, 1 << 17)) + [9 << 20, 3 << 20, (1 << 20) + 1, 0, 1] + [1 << 20]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    os.makedirs(os.path.join(build, "DEBIAN"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("x")
    os.symlink("f000002.bin", os.path.join(build, "d0000", "ln"))
    want = oracle.hashes_yaml(build, tar)
    with Context(flags=0) as c:
        y = c.tree(build, tar)
        assert y == want
        ex = c.stats_ex()
        assert ex["host_bytes"] > 0  # the 9 MiB member at least: 0.2 s alone on the GPU
        assert c.verify(build, y, tar) is None
        victim = os.path.join(build, "d0000", "f000005.bin")
        data = open(victim, "rb").read()
        open(victim, "wb").write(data[:-1] + bytes([data[-1] ^ 1]))
        assert c.verify(build, y, tar) == (4, "d0000/f000005.bin")
        open(victim, "wb").write(data)
        out = os.path.join(str(tmp_path), "o.tar.gz")
        y2, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
        raw = open(out, "rb").read()
        assert hashlib.sha512(raw).digest() == dig and y2 == oracle.hashes_yaml(build, out)
        tf = tarfile.open(fileobj=io.BytesIO(gzip.decompress(raw)))
        members = {m.name: m for m in tf}
        assert "./d0000/f000005.bin" in members and members["./d0000/ln"].issym()
        assert tf.extractfile(members["./d0000/f000005.bin"]).read() == data
    with Context(flags=0, devices=[0, 0]) as c:  # the same through two engines
        assert c.tree(build, tar) == want


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_eight_engines_in_one_process(built_lib, oracle):
    """Row e's in-library form as an 8-GPU node will run it, rehearsed on the one GPU this pool gives a test: devices =
    {0} x 8 (eight engines, copy gather: RCCL needs distinct devices) over BASELINE config 2's own list -- 10 001 x 1 MiB,
    1 250 per engine.  Digests = the single-engine vector = the oracle on a sample; every engine worked; engines on one
    NUMA node hold disjoint CPU slices and their staging sits on the GPU's node; pinned and HBM footprints are what
    3 slots of 256 MiB per engine come to.  (seam: snappy/build.go:517-520.)"""
    import ctypes
    import time
    import torch
    from snappy_amd import Context, _lib, synthetic
    n = 10001
    avail = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) * 1024
    if avail < (40 << 30):
        n = 2001  # a small host: the same shape at a fifth of the size
    sizes = np.full(n, 1 << 20, dtype=np.uint64)
    off, total = synthetic.pack_offsets(sizes)
    with Context(device=0) as c0:
        dev = torch.empty(total, dtype=torch.uint8, device="cuda")
        c0.fill_synthetic_device(dev.data_ptr(), off, sizes, np.arange(n, dtype=np.uint64))
        torch.cuda.synchronize()
        host = dev.cpu().numpy()
        del dev
        torch.cuda.empty_cache()
    ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + int(o) for o in off])
    lens = (ctypes.c_uint64 * n)(*[1 << 20] * n)
    L = _lib.lib()

    def run(c):
        out = ctypes.create_string_buffer(64 * n)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            assert L.snaphash_sha512_buffers(c._h, ptrs, lens, n, out) == 0
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return out.raw, best
    free0, _ = torch.cuda.mem_get_info()
    with Context(devices=[0] * 8, flags=_lib.FLAG_GPU_ONLY) as c:
        got8, t8 = run(c)
        ex = c.stats_ex()
        assert ex["n_devices"] == 8 and ex["gather_kind"] == 2 and ex["host_bytes"] == 0 and ex["gpu_bytes"] == n << 20
        streams = [c.device_stats(i)["streams"] for i in range(8)]
        assert sum(streams) == n and min(streams) >= n // 8 and max(streams) <= n // 8 + 1  # LPT over equal streams
        infos = [c.engine_info(i) for i in range(8)]
        cpus = [set(c.engine_cpus(i)) for i in range(8)]
        free1, _ = torch.cuda.mem_get_info()
        for i, e in enumerate(infos):
            assert e["device"] == 0 and e["fill_threads"] >= 2
            if e["numa_node"] >= 0:
                assert e["staging_node"] in (-1, e["numa_node"])  # pinned staging on the GPU's node
                assert cpus[i], "an engine on a known node has a CPU slice"
            want_slots = 3 if (streams[i] << 20) > 2 * (256 << 20) else 2
            assert want_slots * (256 << 20) <= e["pinned_bytes"] <= want_slots * (256 << 20) + (8 << 20), e
            assert want_slots * (256 << 20) <= e["hbm_bytes"] <= want_slots * (256 << 20) + (16 << 20), e
        for i in range(8):
            for j in range(i):
                assert not (cpus[i] & cpus[j]), "engines %d and %d share CPUs" % (i, j)
        hbm = sum(e["hbm_bytes"] for e in infos)
        assert abs((free0 - free1) - hbm) < (1 << 30), (free0 - free1, hbm)  # what the library says it holds is what the device lost
    with Context(device=0, flags=_lib.FLAG_GPU_ONLY) as c:
        got1, t1 = run(c)
    assert got8 == got1
    for i in [0, 1, n // 2, n - 1] + [int(x) for x in np.random.default_rng(2).integers(0, n, size=12)]:
        assert got8[64 * i:64 * i + 64] == oracle.sha512(host[int(off[i]):int(off[i]) + (1 << 20)].tobytes())
    print("eight engines on one GPU: %d x 1 MiB in %.1f ms (one engine: %.1f ms); pinned %.2f GiB, HBM %.2f GiB; CPU slices %s" %
          (n, t8 * 1e3, t1 * 1e3, sum(e["pinned_bytes"] for e in infos) / 2**30, hbm / 2**30, [len(s) for s in cpus]))


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_no_stream_is_left_for_the_end(built_lib):
    """5 000 equal streams through batches that hold 4 096 floors: every batch must serve a fair rotation -- the streams a
    batch had no room for go first in the next one -- so that no stream's share ever grows beyond the floor.  Rounds 1-3
    served the same leading 4 096 streams every time and hashed the other 904 at the end, alone, in shares of up to 20 x
    the floor (kernels of 6-9 ms behind copies of 4.7: profiles/r04_rank_steps.txt).  Structural, not timed: the engine's
    own batch trace (SNAPHASH_TRACE_BATCHES) is read back from a child process."""
    import re
    import subprocess
    import sys
    code = (
        "import sys, ctypes, hashlib\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from snappy_amd import Context, _lib\n"
        "n, size = 5000, 256 << 10\n"
        "host = np.random.default_rng(4).integers(0, 256, size=n * size, dtype=np.uint8)\n"
        "ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + i * size for i in range(n)])\n"
        "lens = (ctypes.c_uint64 * n)(*[size] * n)\n"
        "out = ctypes.create_string_buffer(64 * n)\n"
        "with Context(staging_bytes=64 << 20, flags=_lib.FLAG_GPU_ONLY) as c:\n"
        "    assert _lib.lib().snaphash_sha512_buffers(c._h, ptrs, lens, n, out) == 0\n"
        "for i in (0, 4095, 4096, 4999):\n"
        "    assert out.raw[64 * i:64 * i + 64] == hashlib.sha512(host[i * size:(i + 1) * size]).digest(), i\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ, SNAPHASH_TRACE_BATCHES="1")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280, env=env)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode(errors="replace")[-800:]
    batches = re.findall(r"batch (\d+): S (\d+), (\d+) segments, (\d+) bytes, largest share (\d+), (\d+) streams left behind", r.stderr.decode(errors="replace"))
    assert len(batches) >= 15, r.stderr.decode(errors="replace")[-400:]
    shares = [int(b[4]) for b in batches]
    segments = [int(b[2]) for b in batches]
    assert max(shares) <= 2 * (16 << 10), shares          # the floor from caller memory is 16 KiB: nobody ever needs more
    # the batches between the ramp (round 5: three eighths of a batch first, 15 % more each time) and the end (the last
    # batch is cut in two): full ones, and all 5 000 streams within any two of them
    s_max = max(int(b[1]) for b in batches)
    steady = [int(b[2]) for b in batches[:-3] if int(b[1]) == s_max]
    assert len(steady) >= 8 and max(segments) <= 4096 + 1 and min(steady) >= 3500, segments
    assert int(batches[-1][5]) == 0


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_descriptor_budget_smaller_than_the_tree(built_lib, oracle, tmp_path):
    """The engine keeps a file's descriptor between the batches the file appears in (FdCache) only within what
    RLIMIT_NOFILE leaves: with a budget far below the number of files the rest is opened segment by segment, as round 3
    did, and nothing changes but the speed.  Run in a child process (the limit is per process)."""
    import subprocess
    import sys
    sizes = [int(x) for x in _ragged_sizes(700, 31, 1 << 18)] + [3 << 20, 1 << 20, 5]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    want = oracle.hashes_yaml(build, tar)
    code = (
        "import resource, sys\n"
        "resource.setrlimit(resource.RLIMIT_NOFILE, (560, min(4096, resource.getrlimit(resource.RLIMIT_NOFILE)[1])))\n"   # budget = 560 - 512 = 48 descriptors for 702 files
        "sys.path.insert(0, %r)\n"
        "from snappy_amd import Context, _lib\n"
        "with Context(staging_bytes=4 << 20, flags=_lib.FLAG_GPU_ONLY | _lib.FLAG_KEEP_RLIMIT) as c:\n"   # 4 MiB staging: every file of any size spans batches
        "    assert resource.getrlimit(resource.RLIMIT_NOFILE)[0] == 560\n"
        "    y = c.tree(%r, %r)\n"
        "    assert c.stats()['launches'] > 8\n"
        "sys.stdout.buffer.write(y)\n" % (ROOT, build, tar))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-800:]
    assert r.stdout == want


def test_engine_info_and_numa_flags(built_lib):
    from snappy_amd import Context, _lib
    with Context(devices=[0, 0]) as c:
        c.sha512_buffers([b"a" * 100000, b"b" * 5])
        for i in range(2):
            e = c.engine_info(i)
            assert e["device"] == 0 and e["fill_threads"] >= 2 and len(e["pci_bus_id"]) >= 7
            assert e["numa_node"] >= -1 and e["staging_node"] >= -1
            if e["numa_node"] >= 0 and e["staging_node"] >= 0:
                assert e["staging_node"] == e["numa_node"]  # the pinned staging memory sits on the GPU's node
    with Context(flags=_lib.FLAG_NO_NUMA | _lib.FLAG_GPU_ONLY) as c:
        assert c.engine_info(0)["numa_node"] == -1 and c.engine_info(0)["n_cpus"] == 0


def test_files_equal_and_tar_producer_on_a_multi_device_ctx(built_lib, oracle, tmp_path):
    """Every entry point accepts the ctx the cgo shim creates (devices = {-1}); exercised here with two engines on
    the one GPU ({0,0}): FilesAreEqual pairs are dealt to the engines, the tar producer and the streaming batch run
    on the first one."""
    import gzip
    import io
    import tarfile
    from snappy_amd import Context
    rng = np.random.default_rng(11)
    d = tmp_path / "cmp"
    d.mkdir()
    pairs, want = [], []
    for i in range(40):
        n = int(rng.integers(0, 300000))
        a = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        same = bool(i % 3)
        b = a if same or n == 0 else a[:n // 2] + bytes([a[n // 2] ^ 1]) + a[n // 2 + 1:]
        pa, pb = d / ("a%d" % i), d / ("b%d" % i)
        pa.write_bytes(a)
        pb.write_bytes(b)
        pairs.append((str(pa), str(pb)))
        want.append(a == b)
    pairs.append((str(d / "a0"), str(d / "missing")))
    want.append(False)
    sizes = [0, 1, 511, 512, 513, 70000, 300000, 12345, 1 << 20]
    build, _ = trees.make_synthetic_tree(str(tmp_path / "t"), sizes)
    out = str(tmp_path / "data.tar.gz")
    with Context(devices=[0, 0], staging_bytes=1 << 19) as c:
        assert c.files_equal(pairs) == want
        assert c.device_stats(0)["bytes_hashed"] > 0 and c.device_stats(1)["bytes_hashed"] > 0  # both engines worked
        y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
    raw = open(out, "rb").read()
    assert hashlib.sha512(raw).digest() == dig and y == oracle.hashes_yaml(build, out)
    tf = tarfile.open(fileobj=io.BytesIO(gzip.decompress(raw)))
    members = {m.name: m for m in tf}
    for dp, _, fs in os.walk(build):
        for f in fs:
            p = os.path.join(dp, f)
            rel = "./" + os.path.relpath(p, build)
            if rel.startswith("./DEBIAN"):
                continue
            assert tf.extractfile(members[rel]).read() == open(p, "rb").read()


def test_tree_with_names_that_need_quoting(built_lib, oracle, tmp_path):
    """Names outside the plain set no longer fail the pass (snappy/build.go:264 marshals any name); parity of the quoted
    forms is UNPINNED (tests/test_yaml_names.py says what is checked instead).  Here: the GPU pass over such a tree
    equals the oracle's, PyYAML reads every name back, Verify accepts its own output, and the fused tar producer no
    longer aborts on them."""
    import yaml
    from snappy_amd import Context
    b = tmp_path / "build"
    (b / "DEBIAN").mkdir(parents=True)
    (b / "my dir").mkdir()
    names = ["1.txt", ".hidden", "foo bar", "icon@2x.png", "a~", "x:y", "x: y", "true", "123", "café", "q'uote", "tab\there",
             "long " + "name " * 20 + "end", "my dir/in ner", "-", "#hash"]
    for i, n in enumerate(names):
        with open(os.path.join(str(b), n), "wb") as f:
            f.write(bytes([i]) * (i * 37))
    tar = tmp_path / "data.tar.gz"
    tar.write_bytes(b"xyz")
    with Context() as c:
        y = c.tree(str(b), str(tar))
        assert y == oracle.hashes_yaml(str(b), str(tar))
        doc = yaml.load(y.decode(), Loader=yaml.BaseLoader)
        assert sorted(f["name"] for f in doc["files"]) == sorted(names + ["my dir"])
        assert c.verify(str(b), y, str(tar)) is None
        out = str(tmp_path / "out.tar.gz")
        y2, _ = c.tar_create(out, str(b), str(b) + "/DEBIAN", with_hashes=True)
        assert y2 == oracle.hashes_yaml(str(b), out)


def test_tree_with_a_directory_that_cannot_be_listed(built_lib, oracle, tmp_path):
    """filepath.Walk calls the callback a second time for a directory whose ReadDir fails, and writeHashes' callback never
    looks at that error (snappy/build.go:228): the directory is recorded twice and the walk goes on.  The GPU pass and the
    oracle agree on that, and Verify accepts the result.  (Needs a non-root user: root lists a mode-000 directory.)"""
    from snappy_amd import Context
    if os.geteuid() == 0:
        pytest.skip("root can list a mode-000 directory; tests/test_host.py covers this through an unprivileged child")
    b = tmp_path / "build"
    (b / "a").mkdir(parents=True)
    (b / "locked" / "inner").mkdir(parents=True)
    (b / "z").mkdir()
    for rel in ("a/f", "locked/inner/g", "z/h"):
        (b / rel).write_bytes(rel.encode() * 1000)
    tar = tmp_path / "data.tar.gz"
    tar.write_bytes(b"tar")
    os.chmod(str(b / "locked"), 0o000)
    try:
        with Context() as c:
            y = c.tree(str(b), str(tar))
            assert y == oracle.hashes_yaml(str(b), str(tar))
            assert y.count(b"- name: locked\n") == 2 and b"inner" not in y
            assert c.verify(str(b), y, str(tar)) is None
    finally:
        os.chmod(str(b / "locked"), 0o755)


@pytest.mark.kernels_only("names the configuration of every Context it makes")
def test_the_plan_is_reported_beside_what_the_call_took_and_the_box_is_calibrated(built_lib, oracle, tmp_path):
    """ABI 5: a ctx that may plan measures its box when it is made (the first engine's H2D rate, what
    one thread copies into pinned staging), corrects both from every staged call, plans with them, and reports the
    plan's modelled makespans beside the measured ones (snaphash_stats_ex).  snaphash_get_plan_model + snaphash_plan_streams
    reproduce the ctx's own plan."""
    import ctypes
    from snappy_amd import Context, _lib
    n = 1536
    blob = np.random.default_rng(6).integers(0, 256, size=(n << 20) + 4096, dtype=np.uint8)
    bufs = [blob[(i << 20) + i % 4096:((i + 1) << 20) + i % 4096] for i in range(n)]
    # (four host threads: 16 cores with eight AVX-512 lanes each would take ALL of a batch this size -- they outrun the link --
    # and there would be no GPU part to compare a prediction with)
    with Context(flags=0, host_threads=4) as c:
        k0 = c.calib()
        assert k0["n_dma"] == 1 and 5e9 < k0["dma"] < 200e9, k0                 # a PCIe link of some generation
        assert k0["n_fill_mem"] == 0 and k0["n_fill_files"] == 0, k0            # fill threads are measured by the calls themselves
        m0 = c.plan_model(False)
        assert abs(m0["gpu_link"] - k0["dma"] * 55.0 / 56.7) < 1e6 and m0["fill_rate"] == 9e9
        assert m0["host_lane_gain_pct"] in (100, 320) and m0["cpus"] == _lib.lib().snaphash_usable_cpus()
        c.sha512_buffers(bufs)  # the first staged call of a ctx also pins its staging buffers (~40 ms for 2 x 256 MiB): not the model's business
        runs = []
        for _ in range(3):      # (the least disturbed of three: four host threads, twelve fill threads and the engine's own share 16 cores)
            got = c.sha512_buffers(bufs)
            runs.append((c.stats_ex(), c.stats()))
        ex, st = min(runs, key=lambda r: r[0]["gpu_ms"])
        assert ex["gpu_bytes"] > 0 and ex["host_bytes"] + ex["gpu_bytes"] == n << 20, (ex, c.plan_model(False), c.calib())
        assert ex["planned_gpu_ms"] > 0 and ex["gpu_ms"] > 0 and ex["hash_ms"] >= ex["gpu_ms"] and ex["plan_ms"] < 20
        assert (ex["planned_host_ms"] > 0) == (ex["host_bytes"] > 0) and ex["planned_threads"] >= ex["host_threads_run"]
        # the prediction is a prediction: within a factor of two on any box this suite has met (bench.py flags 25 %)
        assert 0.5 < ex["gpu_ms"] / ex["planned_gpu_ms"] < 2.0, (ex, c.plan_model(False), c.calib())
        k1 = c.calib()
        assert k1["n_dma"] == 5 and k1["n_fill_mem"] >= 2 and k1["fill_mem"] > 1e9 and k1["n_host"] >= 3  # each call was an observation (fills: not a ctx's first call)
        assert 0.5 < k1["dma"] / k0["dma"] < 2.0                                # and agrees with the probe, roughly
        # the ctx's model through the host-only planner gives the ctx's plan (same streams, same model => same split)
        m1 = c.plan_model(False)
        on_host, r = _lib.plan_streams([1 << 20] * n, **m1)
        got2 = c.sha512_buffers(bufs)
        ex2 = c.stats_ex()
        assert sum(on_host) == ex2["host_streams"] and abs(r["gpu_seconds"] * 1e3 - ex2["planned_gpu_ms"]) < 1e-6
        assert got2 == got
        assert st["bytes_hashed"] == n << 20
    for i in (0, 1, n // 2, n - 1):
        assert got[i] == hashlib.sha512(bufs[i].tobytes()).digest()
    with Context() as c:  # GPU only: no plan, no probe, zeros where a plan would be
        c.sha512_buffers(bufs[:64])
        ex = c.stats_ex()
        assert ex["planned_gpu_ms"] == 0 and ex["planned_threads"] == 0 and ex["gpu_ms"] > 0 and c.calib()["n_dma"] <= 1
    # an ABI 4 caller's shorter snaphash_stats_ex is still filled
    with Context(flags=0) as c:
        c.sha512_buffers([b"x"])
        s = _lib.StatsEx()
        s.struct_size = _lib.StatsEx.planned_gpu_ms.offset
        s.planned_gpu_ms = -1.0
        assert _lib.lib().snaphash_get_stats_ex(c._h, ctypes.byref(s)) == 0
        assert s.host_bytes == 1 and s.planned_gpu_ms == -1.0 and s.struct_size == _lib.StatsEx.planned_gpu_ms.offset


@pytest.mark.kernels_only("runs in a child process that names its own configuration")
def test_host_lanes_and_kept_descriptors_share_one_budget(built_lib, oracle, tmp_path):
    """the host part opens a descriptor per lane (eight a thread) beside the descriptors the staging
    fill keeps between batches; under a low RLIMIT_NOFILE (64 descriptors above what the process holds) with 24 host threads the
    two used to overrun the limit and the call failed with EMFILE where the reference's one-file-at-a-time loop
    succeeds.  Now the call divides what the limit leaves, and an open that still finds no descriptor waits for one."""
    import subprocess
    import sys
    sizes = [int(x) for x in _ragged_sizes(900, 41, 1 << 19)] + [300000] * 400 + [5 << 20, 2 << 20, 7]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    want = oracle.hashes_yaml(build, tar)
    code = (
        "import os, resource, sys\n"
        "sys.path.insert(0, %r)\n"
        "from snappy_amd import Context, _lib\n"
        "ys = []\n"
        "for threads in (24, 2):\n"   # 24: every stream on host threads (24 x 8 lanes want 192 descriptors); 2: a GPU part beside 16 lanes
        "    with Context(staging_bytes=8 << 20, host_threads=threads, flags=_lib.FLAG_KEEP_RLIMIT) as c:\n"
        "        c.sha512_buffers([b'warm' * 100000] * 40)\n"   # the runtime has opened what it opens
        "        soft = len(os.listdir('/proc/self/fd')) + 64\n"   # 64 descriptors above what the process holds
        "        resource.setrlimit(resource.RLIMIT_NOFILE, (soft, resource.getrlimit(resource.RLIMIT_NOFILE)[1]))\n"
        "        ys.append(c.tree(%r, %r))\n"
        "        ex = c.stats_ex()\n"
        "        assert ex['host_bytes'] > 0 and ex['host_threads_run'] <= threads, ex\n"
        "        assert threads == 24 or ex['gpu_bytes'] > 0, ex\n"
        "        paths = [l.split(None, 1)[1] for l in open(%r).read().splitlines()]\n"
        "        d = c.sha512_files(paths)\n"
        "        resource.setrlimit(resource.RLIMIT_NOFILE, (4096, resource.getrlimit(resource.RLIMIT_NOFILE)[1]))\n"
        "assert ys[0] == ys[1]\n"
        "sys.stdout.buffer.write(ys[0])\n" % (ROOT, build, tar, str(tmp_path / "paths.txt")))
    with open(str(tmp_path / "paths.txt"), "w") as f:
        for dp, _, fs in os.walk(build):
            for name in fs:
                f.write("p %s\n" % os.path.join(dp, name))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-800:]
    assert r.stdout == want
