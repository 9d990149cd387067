#!/usr/bin/env python3
"""writeHashes end to end on an on-disk BASELINE config-2 tree (10 000 x 1 MiB + archive):
walk + pread + pinned staging + H2D + kernels + YAML, through snaphash_tree, timed next to
the oracle's serial CPU pass over the same tree, and the two hashes.yaml compared byte for
byte.  usage: tools/e2e_tree.py [nfiles] [bytes each] [dir]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402  (checker + CPU timing, not the product path)
from snappy_amd import Context, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
base = sys.argv[3] if len(sys.argv) > 3 else ("/dev/shm" if os.path.isdir("/dev/shm") else None)
tmp = tempfile.mkdtemp(prefix="snaphash_e2e_", dir=base)
try:
    t0 = time.perf_counter()
    build = os.path.join(tmp, "build")
    os.makedirs(build)
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        oracle.fill_synthetic(size, i).tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    oracle.fill_synthetic(size, n).tofile(tar)
    print("tree of %d x %d B built in %.1f s under %s" % (n, size, time.perf_counter() - t0, tmp), flush=True)
    with Context() as c:
        for rep in range(3):
            t0 = time.perf_counter()
            y_gpu = c.tree(build, tar)
            dt = time.perf_counter() - t0
            st = c.stats()
            print("GPU writeHashes pass %d: %.3f s = %.2f GiB/s (kernel %.1f ms, h2d %.1f ms, %d launches)" %
                  (rep, dt, (n + 1) * size / 2**30 / dt, st["kernel_ms"], st["h2d_ms"], st["launches"]), flush=True)
        for rep in range(2):  # the inverse pass (install-time Verify, row f1) over the same tree
            t0 = time.perf_counter()
            res = c.verify(build, y_gpu, tar)
            dt = time.perf_counter() - t0
            print("GPU Verify pass %d: %.3f s = %.2f GiB/s -> %s" % (rep, dt, (n + 1) * size / 2**30 / dt,
                                                                     "match" if res is None else res), flush=True)
            assert res is None
    t0 = time.perf_counter()
    y_cpu = oracle.hashes_yaml(build, tar)
    dt = time.perf_counter() - t0
    print("CPU oracle pass (1 core, serial like the reference): %.3f s = %.2f GiB/s" % (dt, (n + 1) * size / 2**30 / dt))
    print("hashes.yaml: %d bytes, %d records, GPU == CPU: %s" % (len(y_gpu), y_gpu.count(b"- name: "), y_gpu == y_cpu))
    assert y_gpu == y_cpu
finally:
    shutil.rmtree(tmp, ignore_errors=True)
