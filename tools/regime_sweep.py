#!/usr/bin/env python3
"""Kernel-resident throughput of each kernel variant across stream counts (the evidence
behind SNAPHASH_KERNEL_AUTO and DESIGN.md's regime table).  Streams are equal-length,
content synthetic, resident in HBM.  usage: tools/regime_sweep.py > profiles/...txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context, _lib, synthetic  # noqa: E402

KERNELS = [("wide", _lib.KERNEL_WIDE), ("split", _lib.KERNEL_SPLIT), ("pair", _lib.KERNEL_PAIR)]
CASES = [(256, 4 << 20), (1024, 4 << 20), (4096, 2 << 20), (10001, 1 << 20), (16384, 1 << 20), (32768, 512 << 10),
         (65536, 256 << 10), (131072, 128 << 10), (262144, 64 << 10), (524288, 32 << 10)]
print("%9s %10s | %s" % ("streams", "bytes each", " | ".join("%-22s" % k for k, _ in KERNELS)))
for n, size in CASES:
    lens = np.full(n, size, dtype=np.uint64)
    off, total = synthetic.pack_offsets(lens)
    dev = torch.empty(total, dtype=torch.uint8, device="cuda")
    out = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    row, ref = [], None
    for name, k in KERNELS:
        with Context(kernel=k) as c:
            if ref is None:
                c.fill_synthetic_device(dev.data_ptr(), off, lens, np.arange(n, dtype=np.uint64))
            ms = []
            for rep in range(3):
                c.sha512_device(dev.data_ptr(), off, lens, out.data_ptr())
                c.sync()
                ms.append(c.stats()["kernel_ms"])
            d = out.cpu().numpy().tobytes()
            ref = ref or d
            assert d == ref, "kernels disagree"
            best = min(ms)
            row.append("%8.2f ms %7.1f GB/s" % (best, n * size / best / 1e6))
    print("%9d %10d | %s" % (n, size, " | ".join(row)), flush=True)
    del dev, out
    torch.cuda.empty_cache()
