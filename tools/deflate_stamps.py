#!/usr/bin/env python3
"""Phase breakdown of deflate_chunks_kernel (a build with -DSNAPHASH_DEFLATE_STAMPS prints s_memtime deltas of chunk 37):
make -C snappy_amd/csrc stamps && SNAPHASH_LIB=snappy_amd/variants/libsnaphash_stamps.so python tools/deflate_stamps.py
Without SNAPHASH_LIB: the shipped library, kernel time only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snappy_amd import Context  # noqa: E402

rng = np.random.default_rng(5)
words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
data = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=(64 << 20) // 5 + 16) % 2000)[:64 << 20]
with Context() as c:
    for rep in range(2):
        t0 = time.perf_counter()
        gz = c.gzip_buffer(data)
        dt = time.perf_counter() - t0
        st = c.targz_stats()
        print("gzip_buffer %d MiB: %.1f ms wall, deflate kernels %.1f ms = %.2f GB/s, ratio %.4f" %
              (len(data) >> 20, dt * 1e3, st["deflate_ms"], len(data) / st["deflate_ms"] / 1e6, len(gz) / len(data)), flush=True)
