#!/usr/bin/env python3
"""Kernel-resident A/B of the few-long-streams kernels on BASELINE config 2 (10 001 x 1 MiB): best of six
launches per kernel, HIP-event kernel time from the library's stats.  SNAPHASH_LIB selects another build of
the same ABI (experiments under snappy_amd/variants/).  usage: tools/kernel_ab.py [kernels, e.g. pair,quad]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snappy_amd import Context, _lib, synthetic  # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "pair,quad").split(",")
kern = {"wide": _lib.KERNEL_WIDE, "split": _lib.KERNEL_SPLIT, "pair": _lib.KERNEL_PAIR, "quad": _lib.KERNEL_QUAD}
lens = synthetic.config_sizes("C2")
off, total = synthetic.pack_offsets(lens)
dev = torch.empty(total, dtype=torch.uint8, device="cuda")
out = torch.empty((len(lens), 64), dtype=torch.uint8, device="cuda")
tag = os.path.basename(os.environ.get("SNAPHASH_LIB", "default")).replace("libsnaphash_", "").replace(".so", "")
for name in names:
    with Context(kernel=kern[name]) as c:
        c.fill_synthetic_device(dev.data_ptr(), off, lens, np.arange(len(lens), dtype=np.uint64))
        ms = []
        for rep in range(6):
            c.sha512_device(dev.data_ptr(), off, lens, out.data_ptr())
            c.sync()
            ms.append(c.stats()["kernel_ms"])
        print("%-12s %-6s %.2f ms  %.1f GB/s" % (tag, name, min(ms), float(lens.sum()) / min(ms) / 1e6), flush=True)
