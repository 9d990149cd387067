#!/usr/bin/env python3
"""The library's DEFAULT configuration (ABI 3: streams that would set the makespan all by themselves go to host threads)
next to SNAPHASH_FLAG_GPU_ONLY and to the full planner (host_threads = 16), on the configs where it matters:
BASELINE config 5 (Zipf, 256 MiB head) and config 3 scaled to host memory (100 x 256 MiB), host buffers -> digests.
Digests are compared across the three.  usage: tools/defaults_bench.py"""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from snappy_amd import Context, _lib, synthetic  # noqa: E402


def host_tree(sizes):
    off, total = synthetic.pack_offsets(sizes)
    with Context(device=0) as c0:
        dev = torch.empty(max(total, 16), dtype=torch.uint8, device="cuda")
        c0.fill_synthetic_device(dev.data_ptr(), off, sizes, np.arange(len(sizes), dtype=np.uint64))
        host = dev.cpu().numpy()
    return host, off


def run(host, off, sizes, **kw):
    n = len(sizes)
    ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + int(o) for o in off])
    lens = (ctypes.c_uint64 * n)(*[int(x) for x in sizes])
    out = ctypes.create_string_buffer(64 * n)
    with Context(**kw) as c:
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            rc = _lib.lib().snaphash_sha512_buffers(c._h, ptrs, lens, n, out)
            dt = time.perf_counter() - t0
            assert rc == 0, rc
            if best is None or dt < best[0]:
                best = (dt, c.stats_ex(), c.stats())
    return best, out.raw


for name, sizes in (("C5 (100 000 Zipf files)", synthetic.config_sizes("C5")), ("C3 scaled (100 x 256 MiB)", np.full(100, 256 << 20, dtype=np.uint64))):
    host, off = host_tree(sizes)
    total = int(sizes.sum())
    ref = None
    for label, kw in (("SNAPHASH_FLAG_GPU_ONLY", dict(flags=_lib.FLAG_GPU_ONLY)), ("default (flags 0, host_threads 0 = auto)", dict(flags=0)),
                      ("host_threads = 16 (full planner)", dict(flags=0, host_threads=16))):
        if name.startswith("C3") and label.startswith("SNAPHASH_FLAG_GPU_ONLY"):
            print("%s, %s: skipped (6 s per GiB-stream on the GPU alone: 23.8 s measured in round 2)" % (name, label), flush=True)
            continue
        (b, d) = run(host, off, sizes, **kw)
        ex = b[1]
        print("%s, %.2f GiB, %s: %.3f s = %.2f GiB/s; host %d streams / %.1f MiB (busiest thread %.0f ms), GPU %.1f MiB" % (
            name, total / 2**30, label, b[0], total / 2**30 / b[0], ex["host_streams"], ex["host_bytes"] / 2**20, ex["host_ms"], ex["gpu_bytes"] / 2**20), flush=True)
        if ref is None:
            ref = d
        assert d == ref, "digests differ between configurations"
    head = int(np.argmax(sizes))
    assert ref[64 * head:64 * head + 64] == hashlib.sha512(host[int(off[head]):int(off[head]) + int(sizes[head])].tobytes()).digest()
    del host
print("digests identical across the configurations; the longest stream checked against hashlib")
