#!/usr/bin/env python3
"""Soak of the host entry points: random size distributions (equal, Zipf, few huge + many tiny, empties) through
snaphash_sha512_buffers and snaphash_sha512_files with random staging sizes, hybrid scheduling on and off, one and
two engines; every digest checked against hashlib.  Catches planner and slot-boundary mistakes that fixed-size tests
miss.  usage: tools/soak_host.py [seconds]"""
import hashlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    kind = it % 5
    if kind == 0:
        sizes = np.full(int(rng.integers(1, 400)), int(rng.integers(0, 300000)))
    elif kind == 1:
        n = int(rng.integers(50, 3000))
        sizes = np.maximum(1, (1 << 22) // np.arange(1, n + 1) - rng.integers(0, 100, size=n))
    elif kind == 2:
        sizes = np.concatenate([rng.integers(3 << 20, 12 << 20, size=int(rng.integers(1, 4))), rng.integers(0, 2000, size=int(rng.integers(0, 2000)))])
    elif kind == 3:
        sizes = rng.choice([0, 1, 111, 112, 127, 128, 129, 255, 256, 65535, 65536, 65537], size=int(rng.integers(1, 300)))
    else:
        sizes = rng.integers(0, 1 << int(rng.integers(4, 21)), size=int(rng.integers(1, 1500)))
    sizes = rng.permutation(sizes.astype(np.int64))
    bufs = [rng.integers(0, 256, size=int(s), dtype=np.uint8).tobytes() for s in sizes]
    want = [hashlib.sha512(b).digest() for b in bufs]
    staging = int(rng.choice([1 << 16, (1 << 16) + (1 << 14), 1 << 20, 16 << 20]))
    ht = int(rng.choice([0, 0, 3]))
    devs = [0, 0] if rng.integers(0, 4) == 0 else None
    with Context(staging_bytes=staging, host_threads=ht, devices=devs) as c:
        got = c.sha512_buffers(bufs)
        assert list(got) == want, ("buffers", it, kind, staging, ht, devs)
        if it % 3 == 0:
            tmp = tempfile.mkdtemp(prefix="soak_", dir=base)
            try:
                paths = []
                for k, b in enumerate(bufs):
                    p = os.path.join(tmp, "f%05d" % k)
                    with open(p, "wb") as f:
                        f.write(b)
                    paths.append(p)
                got = c.sha512_files(paths)
                assert list(got) == want, ("files", it, kind, staging, ht, devs)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    it += 1
    if it % 10 == 0:
        print("soak: %d rounds ok (last: kind %d, %d streams, %.1f MiB, staging %d, host_threads %d, engines %s)" % (
            it, kind, len(sizes), sizes.sum() / 2**20, staging, ht, devs or [0]), flush=True)
print("soak: %d rounds, all digests equal hashlib's" % it)
