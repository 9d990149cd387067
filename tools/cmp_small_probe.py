#!/usr/bin/env python3
"""helpers.FilesAreEqual / DirUpdated (row f4) on the sizes its caller has (policy.AppArmorDelta: a few dozen files of a few KiB) and on
larger ones: first call of a ctx and best of five, every byte through the compare kernel (SNAPHASH_FLAG_GPU_ONLY) against the default
configuration (small jobs on host threads, staging halves sized for the job).  usage: tools/cmp_small_probe.py"""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.getcwd())
from snappy_amd import Context, _lib
tmp = tempfile.mkdtemp(prefix="snaphash_cmp_", dir="/dev/shm")
try:
    rng = np.random.default_rng(4)
    for shape, n, sz in (("40 policy files of ~6 KiB", 40, 6000), ("400 files of 64 KiB", 400, 65536), ("64 files of 4 MiB", 64, 4 << 20)):
        da, db = os.path.join(tmp, "a"), os.path.join(tmp, "b")
        os.makedirs(da); os.makedirs(db)
        for i in range(n):
            blob = rng.integers(0, 256, size=sz + int(rng.integers(0, 100)), dtype=np.uint8).tobytes()
            open(os.path.join(da, "f%04d" % i), "wb").write(blob)
            if i % 7 == 3: blob = blob[:-1] + bytes([blob[-1] ^ 1])
            open(os.path.join(db, "f%04d" % i), "wb").write(blob)
        A = [os.path.join(da, "f%04d" % i) for i in range(n)]; B = [os.path.join(db, "f%04d" % i) for i in range(n)]
        want = [0 if i % 7 == 3 else 1 for i in range(n)]
        for flags, tag in ((_lib.FLAG_GPU_ONLY, "GPU only"), (0, "default")):
            with Context(flags=flags) as c:
                t0 = time.perf_counter(); got = c.files_equal(list(zip(A, B))); first = (time.perf_counter() - t0) * 1e3
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter(); got = c.files_equal(list(zip(A, B))); ts.append((time.perf_counter() - t0) * 1e3)
                assert [int(x) for x in got] == want
                print("%-28s %-9s first call %8.2f ms, best %7.3f ms" % (shape, tag, first, min(ts)), flush=True)
        shutil.rmtree(da); shutil.rmtree(db)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
