#!/usr/bin/env python3
"""What ONE rank of an 8-GPU config-4 run does: its LPT shard of the 10 001-file tree (1 250 x 1 MiB) from host memory
through snaphash_sha512_buffers, timed; next to the whole tree on the same GPU.  usage: tools/shard_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snappy_amd import Context, _lib  # noqa: E402

with Context(flags=_lib.FLAG_GPU_ONLY) as c:
    for n in (1250, 2500, 5000, 10001):
        lens = np.full(n, 1 << 20, dtype=np.uint64)
        host = np.random.default_rng(1).integers(0, 256, size=n << 20, dtype=np.uint8)
        ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + (i << 20) for i in range(n)])
        clens = (ctypes.c_uint64 * n)(*[1 << 20] * n)
        out = ctypes.create_string_buffer(64 * n)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            rc = _lib.lib().snaphash_sha512_buffers(c._h, ptrs, clens, n, out)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        st = c.stats()
        print("%5d x 1 MiB: best %.1f ms = %.1f GiB/s (h2d %.1f ms, kernels %.1f ms, %d launches); x8 ranks at this rate: %.0f GiB/s" %
              (n, min(ts) * 1e3, n / 1024 / min(ts), st["h2d_ms"], st["kernel_ms"], st["launches"], 8 * n / 1024 / min(ts)), flush=True)
