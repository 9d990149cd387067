#!/usr/bin/env python3
"""The fused Build pass on SMALL packages (what most snaps are): tar + GPU DEFLATE + archive SHA-512 + per-file SHA-512 + hashes.yaml
over trees of a few MiB to a few hundred, first call and best / median of five, against zlib -9 of the same tar on one core
(the reference's gzip level, clickdeb/deb.go:271) plus hashlib over the files.  usage: tools/build_small_probe.py"""
import os, sys, time, tempfile, shutil, zlib, hashlib, tarfile, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context  # noqa: E402

KiB, MiB = 1 << 10, 1 << 20
rng = np.random.default_rng(21)
words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
text = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=(8 * MiB) // 5 + 16) % 2000)[:8 * MiB]
binary = open(sys.executable, "rb").read()
pool = (text + binary) * 4
SHAPES = {
    "one 1 MiB file": [MiB],
    "snap-like 300 (lognormal, median 20 KiB)": np.minimum(16 * MiB, np.maximum(1, rng.lognormal(np.log(20 * KiB), 2.0, size=300))).astype(np.int64),
    "40 x 1 MiB": [MiB] * 40,
    "snap-like 3000": np.minimum(16 * MiB, np.maximum(1, rng.lognormal(np.log(20 * KiB), 2.0, size=3000))).astype(np.int64),
}
tmp = tempfile.mkdtemp(prefix="snaphash_bsmall_", dir="/dev/shm")
try:
    for shape, sizes in SHAPES.items():
        build = os.path.join(tmp, "t", "build")
        os.makedirs(os.path.join(build, "DEBIAN"))
        for i, sz in enumerate(sizes):
            d = os.path.join(build, "d%03d" % (i // 100)); os.makedirs(d, exist_ok=True)
            off = int(rng.integers(0, len(pool) - int(sz) - 1))
            open(os.path.join(d, "f%05d" % i), "wb").write(pool[off:off + int(sz)])
        out = os.path.join(tmp, "t", "data.tar.gz")
        total = int(np.sum(sizes))
        with Context() as c:
            t0 = time.perf_counter(); c.tar_create(out, build, build + "/DEBIAN", with_hashes=True); first = (time.perf_counter() - t0) * 1e3
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True); ts.append((time.perf_counter() - t0) * 1e3)
            zs = c.targz_stats()
        raw = open(out, "rb").read()
        assert hashlib.sha512(raw).digest() == dig
        tar_bytes = zlib.decompress(raw, 31)
        t0 = time.perf_counter(); z9 = zlib.compress(tar_bytes, 9); tz = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); hashlib.sha512(tar_bytes).digest(); hashlib.sha512(z9).digest(); th = (time.perf_counter() - t0) * 1e3
        ts.sort()
        print("%-42s %7.1f MiB | first call %7.2f ms, best %7.2f, median %7.2f ms | ratio %.4f (zlib -9: %.4f) | zlib -9 + two SHA-512 passes on one core: %8.1f ms = %5.0f x" %
              (shape, total / MiB, first, ts[0], ts[2], len(raw) / len(tar_bytes), len(z9) / len(tar_bytes), tz + th, (tz + th) / ts[2]), flush=True)
        shutil.rmtree(os.path.join(tmp, "t"))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
