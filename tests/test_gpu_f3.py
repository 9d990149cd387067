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
from test_f3_host import sample_inputs, _make_tree, f3  # noqa: F401  (the CPU model of the kernel's format)

pytestmark = pytest.mark.gpu


def test_gpu_deflate_round_trips_and_equals_its_cpu_model(built_lib, f3):  # noqa: F811
    """Every sample: gzip.decompress(GPU output) == input, CRC/ISIZE right -- and the bytes equal the serial
    CPU model of the kernel (same chunking, hash, greedy parse, encoder), so the parallel parse is exact."""
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
            assert st["chunks"] == (len(data) + 16383) // 16384


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
    with pytest.raises(Exception):
        with Context() as c:
            c.tar_create(str(tmp_path / "data.tar.xz"), root)  # "unknown compression extension" for anything but .gz here


def test_fused_build_pass_reads_once_and_matches_writehashes(built_lib, oracle, tmp_path):
    """tar + gzip + archive digest + per-file SHA-512 + hashes.yaml from ONE read of every file (rows f2 + f3):
    the yaml equals what snaphash_tree AND the oracle compute afterwards from the tree and the archive that
    was written; files cross staging pieces (1 MiB staging) so chaining values travel between launches."""
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
        assert c.device_stats(0)["streams"] == len(sizes) + 1  # every regular file, hashed on the GPU out of the tar stream
        ex = c.stats_ex()
        assert ex["host_streams"] == 1 and ex["host_bytes"] == os.path.getsize(out)  # the archive digest: one stream, host core
        assert st["streams"] == len(sizes) + 2 and zs["stored_chunks"] > 0 and zs["stored_chunks"] < zs["chunks"]
        assert c.tree(build, out) == yaml_fused
        assert c.verify(build, yaml_fused, out) is None
    assert oracle.hashes_yaml(build, out) == yaml_fused
    assert hashlib.sha512(open(out, "rb").read()).digest() == digest
    tf = tarfile.open(out, "r:gz")
    for m in tf.getmembers():
        if m.isreg():
            assert tf.extractfile(m).read() == open(os.path.join(build, m.name[2:]), "rb").read(), m.name
