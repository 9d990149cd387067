#!/usr/bin/env python3
"""End-to-end rate (on-disk tree in tmpfs -> hashes.yaml, host buffers -> digests) with one and with two staging
engines on the same GPU (device list {0} vs {0,0}): does a second engine's overlap of pread/memcpy, H2D and kernels
pay?  usage: tools/e2e_engines.py [nfiles] [bytes each]"""
import ctypes
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from snappy_amd import Context, _lib, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
sizes = np.full(n + 1, size, dtype=np.uint64)
off, total = synthetic.pack_offsets(sizes)
with Context(device=0) as c0:
    dev = torch.empty(total, dtype=torch.uint8, device="cuda")
    c0.fill_synthetic_device(dev.data_ptr(), off, sizes, np.arange(n + 1, dtype=np.uint64))
    host = dev.cpu().numpy()
del dev
torch.cuda.empty_cache()
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
tmp = tempfile.mkdtemp(prefix="snaphash_e2e2_", dir=base)
try:
    build = os.path.join(tmp, "build")
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        host[int(off[i]):int(off[i]) + size].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    host[int(off[n]):int(off[n]) + size].tofile(tar)
    ptrs = (ctypes.c_void_p * (n + 1))(*[host.ctypes.data + int(o) for o in off])
    lens = (ctypes.c_uint64 * (n + 1))(*[int(x) for x in sizes])
    out = ctypes.create_string_buffer(64 * (n + 1))
    ref = None
    for devices in ([0], [0, 0], [0, 0, 0]):
        with Context(devices=devices) as c:
            tb = tt = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                assert _lib.lib().snaphash_sha512_buffers(c._h, ptrs, lens, n + 1, out) == 0
                tb = min(tb, time.perf_counter() - t0)
            for rep in range(3):
                t0 = time.perf_counter()
                y = c.tree(build, tar)
                tt = min(tt, time.perf_counter() - t0)
            ref = ref or y
            assert y == ref
            print("engines on GPU 0: %d   host buffers -> digests %.1f GiB/s (%.0f ms)   tree -> hashes.yaml %.1f GiB/s (%.0f ms)" % (
                len(devices), total / 2**30 / tb, tb * 1e3, total / 2**30 / tt, tt * 1e3), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
