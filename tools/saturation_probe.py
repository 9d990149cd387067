#!/usr/bin/env python3
"""The VALU-saturated regime under the profiler: 131 072 streams x 128 KiB (16 GiB resident),
WIDE kernel, a few launches.  Run under rocprofv3 (--kernel-trace --stats, then --pmc ...) to
read the clock (GRBM_GUI_ACTIVE / 8 / duration) and the VALU issue share at saturation."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context, _lib, synthetic  # noqa: E402

n, size = 131072, 128 << 10
lens = np.full(n, size, dtype=np.uint64)
off, total = synthetic.pack_offsets(lens)
dev = torch.empty(total, dtype=torch.uint8, device="cuda")
out = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
with Context(kernel=_lib.KERNEL_WIDE) as c:
    c.fill_synthetic_device(dev.data_ptr(), off, lens, np.arange(n, dtype=np.uint64))
    for rep in range(5):
        c.sha512_device(dev.data_ptr(), off, lens, out.data_ptr())
        c.sync()
        ms = c.stats()["kernel_ms"]
        print("launch %d: %.2f ms = %.1f GB/s" % (rep, ms, n * size / ms / 1e6), flush=True)
