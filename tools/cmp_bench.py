#!/usr/bin/env python3
"""Row f4 measurement: the range-comparison kernel (helpers.FilesAreEqual's byte compare) with both
sides resident in HBM, against its roofline (2 bytes read per byte compared; HBM peak 8 TB/s), and
the oracle's serial CPU pass (cmp.go's 16 KiB loop) over files of the same shape in tmpfs.
usage: tools/cmp_bench.py [pairs] [bytes each]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
lens = np.full(n, size, dtype=np.uint64)
off, total = synthetic.pack_offsets(lens)
with Context() as c:
    a = torch.empty(total, dtype=torch.uint8, device="cuda")
    c.fill_synthetic_device(a.data_ptr(), off, lens, np.arange(n, dtype=np.uint64))
    b = a.clone()
    b[int(off[n // 2]) + size - 1] ^= 1  # exactly one differing pair, in its last byte
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ms = []
    for rep in range(8):
        c.ranges_equal_device(a.data_ptr(), off, b.data_ptr(), off, lens, out.data_ptr())
        c.sync()
        ms.append(c.stats()["kernel_ms"])
    res = out.cpu().numpy()
    assert res.sum() == n - 1 and res[n // 2] == 0
    best = min(ms[2:])
    gbs = 2.0 * n * size / best / 1e6
    print("GPU ranges_equal: %d pairs x %d B, kernel %.3f ms, %.1f GB/s read (2 B per byte compared) = %.3f of the 8 TB/s HBM roofline"
          % (n, size, best, gbs, gbs / 8000.0), flush=True)
# CPU: the oracle's restatement of cmp.go over real files (serial, 1 core, like the reference)
from oracle import oracle  # noqa: E402  (checker/baseline only)
tmp = tempfile.mkdtemp(prefix="snaphash_cmp_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    m = min(n, 2048)
    for i in range(m):
        d = oracle.fill_synthetic(size, i)
        d.tofile(os.path.join(tmp, "a%d" % i))
        d.tofile(os.path.join(tmp, "b%d" % i))
    t0 = time.perf_counter()
    ok = sum(oracle.files_equal(os.path.join(tmp, "a%d" % i), os.path.join(tmp, "b%d" % i)) for i in range(m))
    dt = time.perf_counter() - t0
    assert ok == m
    print("CPU oracle FilesAreEqual: %d pairs x %d B in %.2f s = %.2f GB/s read, 1 core" % (m, size, dt, 2.0 * m * size / dt / 1e9))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
