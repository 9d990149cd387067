#!/usr/bin/env python3
"""Where the DEFLATE kernel's time goes inside the fused pass: the same 2 GiB text tree through tar_create with and
without the per-file hashes (no SHA-512 kernels beside the compressor), and through gzip_buffer (no tree at all).
usage: tools/targz_probe.py [MiB]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(5)
words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
block = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=(64 << 20) // 5 + 16) % 2000)[:64 << 20]
tmp = tempfile.mkdtemp(prefix="snaphash_probe_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    build = os.path.join(tmp, "build")
    os.makedirs(build)
    for i in range(mib):
        off = (i * 1048576) % (len(block) - 1048576)
        open(os.path.join(build, "f%05d" % i), "wb").write(block[off:off + 1048576])
    with Context() as c:
        for with_hashes in (True, False, True, False):
            t0 = time.perf_counter()
            c.tar_create(os.path.join(tmp, "data.tar.gz"), build, with_hashes=with_hashes)
            dt = time.perf_counter() - t0
            st = c.targz_stats()
            print("tar_create with_hashes=%-5s %.3f s; deflate kernels %.0f ms (%.1f ms per 64 MiB), sha kernels %.0f ms, fill %.0f ms" %
                  (with_hashes, dt, st["deflate_ms"], st["deflate_ms"] / (mib / 64), st.get("sha_ms", 0), st["fill_ms"]), flush=True)
        data = block * (256 // 64)
        for rep in range(2):
            t0 = time.perf_counter()
            c.gzip_buffer(data)
            dt = time.perf_counter() - t0
            st = c.targz_stats()
            print("gzip_buffer 256 MiB: %.3f s; deflate kernels %.1f ms (%.1f ms per 64 MiB)" % (dt, st["deflate_ms"], st["deflate_ms"] / 4), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
