#!/usr/bin/env python3
"""One rank's C4 shard (1 250 x 1 MiB from host memory), the GPU side batch by batch (SNAPHASH_TRACE_EVENTS) and the host side
in sums (SNAPHASH_TRACE_TREE).  usage: SNAPHASH_TRACE_EVENTS=1 SNAPHASH_TRACE_TREE=1 tools/shard_trace.py [n]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snappy_amd import Context, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
files = len(sys.argv) > 2 and sys.argv[2] == "files"
with Context(flags=_lib.FLAG_GPU_ONLY) as c:
    host = np.random.default_rng(1).integers(0, 256, size=n << 20, dtype=np.uint8)
    if files:  # the same shard as files on tmpfs (what a rank of bench.py --gpus 8 reads): snaphash_sha512_files
        import shutil, tempfile
        tmp = tempfile.mkdtemp(prefix="snaphash_shard_", dir="/dev/shm")
        paths = []
        for i in range(n):
            p = os.path.join(tmp, "f%05d.bin" % i)
            host[i << 20:(i + 1) << 20].tofile(p)
            paths.append(p)
        try:
            for rep in range(4):
                sys.stderr.write("---- files rep %d\n" % rep)
                t0 = time.perf_counter()
                c.sha512_files(paths)
                sys.stderr.write("wall %.2f ms\n" % ((time.perf_counter() - t0) * 1e3))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        sys.exit(0)
    ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + (i << 20) for i in range(n)])
    clens = (ctypes.c_uint64 * n)(*[1 << 20] * n)
    out = ctypes.create_string_buffer(64 * n)
    for rep in range(4):
        sys.stderr.write("---- rep %d\n" % rep)
        t0 = time.perf_counter()
        assert _lib.lib().snaphash_sha512_buffers(c._h, ptrs, clens, n, out) == 0
        sys.stderr.write("wall %.2f ms\n" % ((time.perf_counter() - t0) * 1e3))
