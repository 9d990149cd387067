#!/usr/bin/env python3
"""The few-huge-streams regime, measured: BASELINE config 5 (Zipf, 256 MiB head) from
host buffers with every byte on the GPU (default) and with opt-in hybrid scheduling (host_threads = 16: the
streams whose single-stream GPU time would set the makespan finish on host threads running the library's own
SHA-512); and a package whose own archive is as large as its tree through snaphash_tree, three ways: archive on
the GPU (default), hybrid, and the archive digest supplied by the caller (snaphash_tree_ex).
usage: tools/hybrid_bench.py [archive MiB]"""
import ctypes
import hashlib
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

arch_mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def host_tree(sizes):
    off, total = synthetic.pack_offsets(sizes)
    with Context(device=0) as c0:
        dev = torch.empty(max(total, 16), dtype=torch.uint8, device="cuda")
        c0.fill_synthetic_device(dev.data_ptr(), off, sizes, np.arange(len(sizes), dtype=np.uint64))
        host = dev.cpu().numpy()
    return host, off


def run_buffers(host, off, sizes, **kw):
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


sizes = synthetic.config_sizes("C5")
host, off = host_tree(sizes)
total = int(sizes.sum())
(b0, d0) = run_buffers(host, off, sizes)
print("C5 (100 000 Zipf files, %.2f GiB), host buffers -> digests, GPU only: %.3f s = %.2f GiB/s (kernel %.0f ms)" % (
    total / 2**30, b0[0], total / 2**30 / b0[0], b0[2]["kernel_ms"]), flush=True)
for T in (4, 16):
    (b1, d1) = run_buffers(host, off, sizes, host_threads=T)
    ex = b1[1]
    print("C5, hybrid host_threads=%d: %.3f s = %.2f GiB/s; host: %d streams / %.1f MiB (busiest thread %.0f ms), GPU: %.1f MiB; digests identical: %s" % (
        T, b1[0], total / 2**30 / b1[0], ex["host_streams"], ex["host_bytes"] / 2**20, ex["host_ms"], ex["gpu_bytes"] / 2**20, d0 == d1), flush=True)
    assert d0 == d1
head = int(np.argmax(sizes))
assert d0[64 * head:64 * head + 64] == hashlib.sha512(host[int(off[head]):int(off[head]) + int(sizes[head])].tobytes()).digest()
del host

# config 3 scaled to host memory: 100 x 256 MiB from host buffers
sizes3 = np.full(100, 256 << 20, dtype=np.uint64)
host, off = host_tree(sizes3)
total = int(sizes3.sum())
(b0, d0) = run_buffers(host, off, sizes3)
print("C3 scaled (100 x 256 MiB = 25 GiB), host buffers -> digests, GPU only: %.2f s = %.2f GiB/s" % (b0[0], total / 2**30 / b0[0]), flush=True)
(b1, d1) = run_buffers(host, off, sizes3, host_threads=16)
ex = b1[1]
print("C3 scaled, hybrid host_threads=16: %.2f s = %.2f GiB/s; host: %d streams / %.1f GiB (busiest thread %.0f ms), GPU: %.1f GiB; digests identical: %s" % (
    b1[0], total / 2**30 / b1[0], ex["host_streams"], ex["host_bytes"] / 2**30, ex["host_ms"], ex["gpu_bytes"] / 2**30, d0 == d1), flush=True)
assert d0 == d1
del host

# a package whose data.tar.gz is as large as its tree (build.go:222 hashes it as ONE stream)
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
tmp = tempfile.mkdtemp(prefix="snaphash_hyb_", dir=base)
try:
    build = os.path.join(tmp, "build")
    n = arch_mib
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "wb") as f:
            f.write(synthetic.file_bytes(1 << 20, i))
    tar = os.path.join(tmp, "data.tar.gz")
    with open(tar, "wb") as f:
        for i in range(n):
            f.write(synthetic.file_bytes(1 << 20, 100000 + i))
    tot = (2 * n) << 20
    with Context() as c:
        t0 = time.perf_counter(); y0 = c.tree(build, tar); dt = time.perf_counter() - t0
        print("tree of %d x 1 MiB + a %d MiB archive, archive on the GPU (default): %.2f s = %.2f GiB/s" % (n, n, dt, tot / 2**30 / dt), flush=True)
    with Context(host_threads=16) as c:
        t0 = time.perf_counter(); y1 = c.tree(build, tar); dt = time.perf_counter() - t0
        ex = c.stats_ex()
        print("same, hybrid host_threads=16: %.2f s = %.2f GiB/s (host %d stream(s), %.0f MiB)" % (dt, tot / 2**30 / dt, ex["host_streams"], ex["host_bytes"] / 2**20), flush=True)
    with Context() as c:
        t0 = time.perf_counter()
        arch = hashlib.sha512(open(tar, "rb").read()).digest()  # the caller's own CPU hash of the archive (Go: crypto/sha512)
        t_a = time.perf_counter() - t0
        t0 = time.perf_counter(); y2 = c.tree_ex(build, None, arch); dt = time.perf_counter() - t0
        print("same, archive digest supplied by the caller (snaphash_tree_ex): tree pass %.2f s; the caller's hash of the archive %.2f s beside it" % (dt, t_a), flush=True)
    assert y0 == y1 == y2
    print("hashes.yaml identical in all three")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
