#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry point (never bench.py's `value`): C2-shaped
content in host memory -> snaphash_sha512_buffers (pinned staging, double-buffered H2D,
chunked segments) -> digests on the host.  usage: tools/e2e_bench.py [nfiles] [MiB each]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10001
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1
size = mib << 20
host = np.random.default_rng(0).integers(0, 256, size=n * size, dtype=np.uint8)
ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + i * size for i in range(n)])
lens = (ctypes.c_uint64 * n)(*([size] * n))
out = ctypes.create_string_buffer(64 * n)
with Context() as c:
    for rep in range(3):
        t0 = time.perf_counter()
        rc = _lib.lib().snaphash_sha512_buffers(c._h, ptrs, lens, n, out)
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        st = c.stats()
        print("rep %d: %.3f s  %.2f GiB/s end-to-end  (kernel %.1f ms, h2d %.1f ms, launches %d)" %
              (rep, dt, n * size / 2**30 / dt, st["kernel_ms"], st["h2d_ms"], st["launches"]), flush=True)
import hashlib
assert out.raw[:64] == hashlib.sha512(host[:size].tobytes()).digest()
print("first digest bit-exact vs hashlib")
