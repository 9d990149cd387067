"""The N > 1 path on CPU: world_size-2 gloo ranks run the shard plan + digest
gather (snappy_amd/sharded.py).  Hashing itself needs a GPU, so each rank's
local digests come from the oracle here -- the test covers the partitioning and
the collective, which are device-independent."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from snappy_amd.sharded import ShardPlan, gather_digests
        plan = ShardPlan(sizes, world)
        mine = plan.members(rank)
        slab = torch.zeros((plan.kmax, 64), dtype=torch.uint8)
        for k, i in enumerate(mine):
            d = oracle.sha512(oracle.fill_synthetic(int(sizes[i]), int(i)).tobytes())
            slab[k] = torch.frombuffer(bytearray(d), dtype=torch.uint8)
        full = gather_digests(slab, plan)
        q.put((rank, full.numpy().tobytes(), [int(c) for c in plan.counts]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_and_gather_gloo(world, oracle):
    rng = np.random.default_rng(5)
    sizes = np.concatenate([rng.integers(0, 3000, size=41), [0, 128, 100000]]).astype(np.uint64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = b"".join(oracle.sha512(oracle.fill_synthetic(int(n), i).tobytes()) for i, n in enumerate(sizes))
    for rank, blob, counts in res:
        assert blob == want, rank               # every rank holds the full vector, in walk order
        assert sum(counts) == len(sizes) and min(counts) > 0


def test_plan_is_balanced_and_deterministic(built_lib):
    from snappy_amd import synthetic
    from snappy_amd.sharded import ShardPlan
    sizes = np.tile(synthetic.config_sizes("C2"), 8)
    p = ShardPlan(sizes, 8)
    assert set(p.counts) == {10001}
    assert sorted(p.row_of.tolist()) == list(range(8 * 10001))
    z = ShardPlan(synthetic.zipf_sizes(20000), 8)
    loads = np.array([synthetic.zipf_sizes(20000)[z.members(r)].sum() for r in range(8)], dtype=np.float64)
    # the 256 MiB head file bounds the makespan from below; LPT stays within one head file of the mean
    assert loads.max() <= max(loads.mean() + (1 << 28), 1 << 28)


def _tree_worker(rank, world, port, build, tar, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hashlib
        from snappy_amd.sharded import ShardedTree
        with ShardedTree(build, tar, rank, world) as st:
            slab = np.zeros((st.rows, 64), dtype=np.uint8)
            for k, p in enumerate(st.paths()):  # no GPU here: this rank's digests from hashlib, into its slab rows
                slab[k] = np.frombuffer(hashlib.sha512(open(p, "rb").read()).digest(), dtype=np.uint8)
            y = st.emit(st.gather(slab))
            q.put((rank, y, st.count, st.rows, st.streams, st.bytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_tree_plan_gather_emit_gloo(world, oracle, built_lib, tmp_path):
    """ABI 4 snaphash_shard_plan / _emit (host-only halves of the one-process-per-GPU pass): every rank walks the tree,
    takes its LPT share, the slabs are all-gathered over gloo, and EVERY rank writes the oracle's hashes.yaml."""
    import trees
    rng = np.random.default_rng(8)
    sizes = [int(x) for x in rng.integers(0, 5000, size=57)] + [0, 128, 300000, 70000, 12345]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    os.makedirs(os.path.join(build, "DEBIAN"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("x")
    os.symlink("f000001.bin", os.path.join(build, "d0000", "lnk"))
    want = oracle.hashes_yaml(build, tar)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tree_worker, args=(r, world, port, build, tar, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[2] for r in res) == len(sizes) == res[0][4]  # every stream (files + the archive) has exactly one owner
    assert max(r[2] for r in res) == res[0][3]
    loads = [r[5] for r in res]
    assert max(loads) - min(loads) <= 300000  # LPT: within the largest member of each other
    for rank, y, *_ in res:
        assert y == want, rank


def test_sharded_tree_world_one_equals_the_plain_emitter(built_lib, oracle, tmp_path):
    import hashlib
    import trees
    from snappy_amd.sharded import ShardedTree
    build, tar = trees.make_simple_tree(str(tmp_path))
    with ShardedTree(build, tar, 0, 1) as st:
        slab = np.zeros((st.rows, 64), dtype=np.uint8)
        for k, p in enumerate(st.paths()):
            slab[k] = np.frombuffer(hashlib.sha512(open(p, "rb").read()).digest(), dtype=np.uint8)
        assert st.emit(st.gather(slab)) == open(os.path.join(ROOT, "tests", "golden", "hashes_simple.yaml"), "rb").read()
    from snappy_amd import _lib
    with pytest.raises(_lib.SnaphashError):
        ShardedTree(build, tar + ".missing", 0, 1)
    with pytest.raises(_lib.SnaphashError):
        ShardedTree(build, tar, 2, 2)


@pytest.mark.parametrize("world", [1, 3, 8])
def test_sharded_tree_edge_shapes_without_a_collective(world, built_lib, oracle, tmp_path):
    """More ranks than streams, an empty tree (files: []), a tree of directories and symlinks only, zero-length files:
    every rank's plan is made in this one process, the slabs are laid rank-major by hand, and any rank's emit must write
    the oracle's hashes.yaml.  (ABI 4 snaphash_shard_plan / _emit are host-only.)"""
    import hashlib
    from snappy_amd.sharded import ShardedTree

    def run(build, tar):
        sts = [ShardedTree(build, tar, r, world) for r in range(world)]
        try:
            rows = sts[0].rows
            assert all(st.rows == rows and st.streams == sts[0].streams for st in sts)
            assert sum(st.count for st in sts) == sts[0].streams
            slabs = np.zeros((world * rows, 64), dtype=np.uint8)
            for r, st in enumerate(sts):
                for k, p in enumerate(st.paths()):
                    slabs[r * rows + k] = np.frombuffer(hashlib.sha512(open(p, "rb").read()).digest(), dtype=np.uint8)
            want = oracle.hashes_yaml(build, tar)
            for st in sts:
                assert st.emit(slabs) == want
        finally:
            for st in sts:
                st.close()

    tar = tmp_path / "data.tar.gz"
    tar.write_bytes(b"the archive")
    empty = tmp_path / "empty"
    empty.mkdir()
    run(str(empty), str(tar))                       # the archive is the only stream: files: []
    only_debian = tmp_path / "onlydeb"
    (only_debian / "DEBIAN").mkdir(parents=True)
    (only_debian / "DEBIAN" / "control").write_text("x")
    run(str(only_debian), str(tar))                 # everything skipped by the /DEBIAN prefix rule
    shapes = tmp_path / "shapes"
    (shapes / "a" / "b").mkdir(parents=True)
    os.symlink("nowhere", str(shapes / "a" / "dangling"))
    (shapes / "a" / "zero").write_bytes(b"")
    (shapes / "a-b").write_bytes(b"x")              # Walk order: a, a/..., a-b -- not a global sort of the paths
    (shapes / "DEBIANfoo").write_bytes(b"skipped")  # string prefix, not path component (build.go:229)
    run(str(shapes), str(tar))


def _disagree_worker(rank, world, port, builds, tars, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        import hashlib
        from snappy_amd import _lib
        from snappy_amd.sharded import ShardedTree
        try:
            with ShardedTree(builds[rank], tars[rank], rank, world) as st:
                slab = np.zeros((max(st.rows, 1), 64), dtype=np.uint8)
                for k, p in enumerate(st.paths()):
                    slab[k] = np.frombuffer(hashlib.sha512(open(p, "rb").read()).digest(), dtype=np.uint8)
                st.emit(st.gather(slab))
            q.put((rank, "no error", 0))
        except _lib.SnaphashError as e:
            q.put((rank, str(e), e.code))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["another tree", "one rank's plan fails"])
def test_ranks_that_disagree_all_raise_before_the_gather(case, built_lib, tmp_path):
    """a tree that changed between the ranks' walks, or a rank that failed, must not end in mismatched slabs
    or in ranks blocked in the collective: ShardedTree.gather first all-gathers (rc, streams, rows, fingerprint of the
    plan) and EVERY rank raises -- as the reference's serial loop returns its first error (snappy/build.go:242-244)."""
    import shutil
    import trees
    from snappy_amd import _lib
    build, tar = trees.make_synthetic_tree(str(tmp_path / "a"), [100, 2000, 30000, 7, 512])
    other = str(tmp_path / "b" / "build")
    shutil.copytree(build, other)
    if case == "another tree":
        open(os.path.join(other, "d0000", "f000001.bin"), "ab").write(b"one more byte")  # same names, another size
        builds, tars = [build, other], [tar, tar]
    else:
        builds, tars = [build, build], [tar, tar + ".gone"]  # rank 1: build.go:222's missing archive
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_disagree_worker, args=(r, 2, port, builds, tars, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, msg, code in res:
        assert code == (_lib.EMISMATCH if case == "another tree" else _lib.EIO), (rank, msg)
    assert ("another tree" in res[0][1]) if case == "another tree" else ("rank 1 failed in snaphash_shard_plan" in res[0][1])


def test_shard_fingerprint_and_local_ranks(built_lib, tmp_path):
    import trees
    from snappy_amd import _lib
    from snappy_amd.sharded import ShardedTree
    build, tar = trees.make_synthetic_tree(str(tmp_path), [100, 2000, 30000, 7, 512])
    with ShardedTree(build, tar, 0, 2) as a, ShardedTree(build, tar, 1, 2, local_ranks=2) as b, ShardedTree(build, tar, 0, 3) as c:
        assert a.fingerprint == b.fingerprint != 0        # the same plan on every rank
        assert c.fingerprint != a.fingerprint             # another world is another plan
        assert _lib.lib().snaphash_shard_set_local_ranks(a._h, 3) == _lib.EINVAL  # more ranks on the node than in the job
        assert _lib.lib().snaphash_shard_set_local_ranks(a._h, 0) == 0
    os.chmod(os.path.join(build, "d0000", "f000000.bin"), 0o600)
    with ShardedTree(build, tar, 0, 2) as d:
        assert d.fingerprint != a.fingerprint             # a mode is part of what hashes.yaml says


def test_bench_starts_its_own_ranks_and_fails_loudly_without_gpus():
    """`python3 bench.py --gpus N` with no launcher around it starts its N ranks itself, as child
    processes, before the parent has imported torch or the library.  Here (no GPU) every rank refuses to run; what is
    checked is the launcher: the parent relays the failure, ends the other ranks, exits non-zero and prints no JSON line.
    (On the GPU box the same command prints the line: profiles/r05_bench_2ranks_same_gpu.json.)"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, env=env)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0 and r.stdout.strip() == b""
    assert "the other ranks were ended" in err and "no CPU fallback exists" in err, err[-600:]
    # the parent itself never loaded torch: it is the ranks that say so
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def self_launch("):src.index("def main(")]
    import re
    loads = re.compile(r"^\s*(import|from)\s+(torch|snappy_amd|oracle)\b", re.M)
    assert not loads.search(body)
    head = src[src.index("def main("):src.index("sys.exit(self_launch(args))")]
    assert not loads.search(head)


def test_large_tree_yaml_written_ahead_of_its_digests(built_lib, oracle, tmp_path):
    """Round 5: for a tree of 2 048 records or more hashes.yaml is written from the end of the plan on, by a background
    thread, with zeros where the digests go (hostpass.cpp emit_yaml_skeleton); snaphash_shard_emit fills the digests in.
    The bytes must be the oracle's -- odd names (quoted, folded) and empty files included -- and a second emit with other
    slabs must not see the first one's digests."""
    import hashlib
    from snappy_amd.sharded import ShardedTree
    rng = np.random.default_rng(12)
    build = tmp_path / "build"
    for d in range(30):
        (build / ("dir %02d" % d if d % 7 == 0 else "d%02d" % d)).mkdir(parents=True)
    names = []
    for i in range(2600):
        d = i % 30
        dn = "dir %02d" % d if d % 7 == 0 else "d%02d" % d
        fn = {0: "f%04d" % i, 1: "true", 2: "1e3", 3: "a: b %d" % i, 4: "x" * 90 + str(i)}[i % 5 if i % 97 == 0 else 0]
        p = build / dn / fn
        if p.exists():
            fn = "f%04d" % i
            p = build / dn / fn
        p.write_bytes(rng.integers(0, 256, size=int(rng.integers(0, 300)), dtype=np.uint8).tobytes())
        names.append(str(p))
    tar = tmp_path / "data.tar.gz"
    tar.write_bytes(b"archive bytes")
    want = oracle.hashes_yaml(str(build), str(tar))
    with ShardedTree(str(build), str(tar), 0, 1) as st:
        assert st.streams == 2601
        slab = np.zeros((st.rows, 64), dtype=np.uint8)
        for k, p in enumerate(st.paths()):
            slab[k] = np.frombuffer(hashlib.sha512(open(p, "rb").read()).digest(), dtype=np.uint8)
        assert st.emit(slab) == want
        other = st.emit(np.zeros_like(slab))
        assert other != want and len(other) == len(want) and other.count(b"0" * 128) == 2601
        assert st.emit(slab) == want


def _shared_walk_worker(rank, world, port, builds, tar, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        import hashlib
        from snappy_amd import _lib
        from snappy_amd.sharded import ShardedTree
        try:
            with ShardedTree(builds[rank], tar, rank, world, share_walk={}) as st:
                fp, count, rows = st.fingerprint, st.count, st.rows
                slab = np.zeros((max(st.rows, 1), 64), dtype=np.uint8)
                for k, p in enumerate(st.paths()):
                    slab[k] = np.frombuffer(hashlib.sha512(open(p, "rb").read()).digest(), dtype=np.uint8)
                y = st.emit(st.gather(slab))
            q.put((rank, "ok", y, fp, count, rows))
        except _lib.SnaphashError as e:
            q.put((rank, "error", str(e), e.code, 0, 0))
    finally:
        dist.destroy_process_group()


def _shared_walk_tree(tmp_path):
    import trees
    rng = np.random.default_rng(31)
    sizes = [int(x) for x in rng.integers(0, 4000, size=230)] + [0, 128, 90000, 1]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    os.makedirs(os.path.join(build, "DEBIAN", "deep"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("x")
    open(os.path.join(build, "DEBIAN", "deep", "y"), "w").write("y")
    open(os.path.join(build, "DEBIANfoo"), "w").write("skipped: a string prefix, not a path component")
    open(os.path.join(build, "a-file-in-the-root"), "w").write("top")
    os.makedirs(os.path.join(build, "zz", "sub", "subsub"))
    open(os.path.join(build, "zz", "sub", "subsub", "leaf"), "w").write("leaf")
    os.makedirs(os.path.join(build, "zz-empty"))
    os.symlink("sub", os.path.join(build, "zz", "link-to-dir"))
    os.symlink("nowhere", os.path.join(build, "dangling"))
    return build, tar


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_share_the_walk_gloo(world, oracle, built_lib, tmp_path):
    """ABI 5 snaphash_shard_list / _plan_from: rank r walks the subtrees of every world-th entry of the root, the listings are
    all-gathered (gloo here, RCCL in bench.py), and every rank rebuilds the records of the WHOLE tree in filepath.Walk's
    order: the plan's fingerprint is the one snaphash_shard_plan gives for a full walk, hashes.yaml the oracle's."""
    from snappy_amd.sharded import ShardedTree
    build, tar = _shared_walk_tree(tmp_path)
    want = oracle.hashes_yaml(build, tar)
    full = [ShardedTree(build, tar, r, world) for r in range(world)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_walk_worker, args=(r, world, port, [build] * world, tar, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, status, y, fp, count, rows in res:
        assert status == "ok", (rank, y)
        assert y == want, rank
        assert (fp, count, rows) == (full[rank].fingerprint, full[rank].count, full[rank].rows), rank
    for st in full:
        st.close()


@pytest.mark.parametrize("case", ["one rank cannot list", "the ranks see different roots"])
def test_shared_walk_failures_reach_every_rank(case, built_lib, tmp_path):
    import shutil
    from snappy_amd import _lib
    build, tar = _shared_walk_tree(tmp_path / "a")
    if case == "one rank cannot list":
        builds = [build, build + "-gone"]
    else:
        other = str(tmp_path / "b" / "build")
        shutil.copytree(build, other, symlinks=True)
        open(os.path.join(other, "one-more-in-the-root"), "w").write("x")
        builds = [build, other]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_walk_worker, args=(r, 2, port, builds, tar, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, status, msg, code, *_ in res:
        assert status == "error", (rank, msg)
        assert code == (_lib.EIO if case == "one rank cannot list" else _lib.EMISMATCH), (rank, msg, code)


def test_shard_plan_from_rejects_malformed_listings(built_lib, tmp_path):
    """The listings come from peer ranks, but a truncated or foreign buffer must come back as an error, never be read past."""
    import ctypes
    from snappy_amd import _lib
    build, tar = _shared_walk_tree(tmp_path)
    L = _lib.lib()
    blobs = []
    for r in range(2):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        assert L.snaphash_shard_list(build.encode(), r, 2, ctypes.byref(p), ctypes.byref(n)) == 0
        blobs.append(ctypes.string_at(p, n.value))
        L.snaphash_free(p)

    def plan(b0, b1):
        keep = [ctypes.create_string_buffer(b0, len(b0)), ctypes.create_string_buffer(b1, len(b1))]
        ptrs = (ctypes.c_void_p * 2)(*[ctypes.addressof(k) for k in keep])
        sizes = (ctypes.c_size_t * 2)(len(b0), len(b1))
        h = ctypes.c_void_p()
        rc = L.snaphash_shard_plan_from(build.encode(), tar.encode(), 0, 2, ptrs, sizes, ctypes.byref(h))
        if rc == 0:
            L.snaphash_shard_free(h)
        return rc
    assert plan(blobs[0], blobs[1]) == 0
    assert plan(blobs[1], blobs[0]) == _lib.EPARSE                       # a blob in another rank's place
    assert plan(blobs[0], blobs[1][:-3]) == _lib.EPARSE                  # cut short
    assert plan(blobs[0], blobs[1] + b"x") == _lib.EPARSE                # trailing bytes
    assert plan(blobs[0], b"") == _lib.EPARSE
    assert plan(b"\\xff" * len(blobs[0]), blobs[1]) == _lib.EPARSE
    for cut in range(0, len(blobs[0]), max(1, len(blobs[0]) // 97)):      # every kind of truncation
        assert plan(blobs[0][:cut], blobs[1]) in (_lib.EPARSE, _lib.EMISMATCH)
