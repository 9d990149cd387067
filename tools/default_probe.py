#!/usr/bin/env python3
"""The planner's mixed regime measured: the C2 tree (files on tmpfs) and the same bytes from host memory, GPU only against
host_threads = 1..8 beside the GPU part (explicit counts: the planner then balances with exactly that many).
usage: tools/default_probe.py [n=10000]"""
import ctypes, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench  # noqa: E402
from snappy_amd import Context, _lib, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
print(bench.bind_to_gpu_node(0))
tmp = tempfile.mkdtemp(prefix="snaphash_dp_", dir="/dev/shm")
try:
    build = os.path.join(tmp, "build")
    host = np.random.default_rng(3).integers(0, 256, size=(n + 1) << 20, dtype=np.uint8)
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        host[i << 20:(i + 1) << 20].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    host[n << 20:].tofile(tar)
    ptrs = (ctypes.c_void_p * (n + 1))(*[host.ctypes.data + (i << 20) for i in range(n + 1)])
    lens = (ctypes.c_uint64 * (n + 1))(*[1 << 20] * (n + 1))
    out = ctypes.create_string_buffer(64 * (n + 1))
    total = (n + 1) << 20
    for label, kw in [("GPU only", dict(flags=_lib.FLAG_GPU_ONLY)), ("default (auto)", dict(flags=0))] + [("host_threads=%d" % t, dict(flags=0, host_threads=t)) for t in (1, 2, 3, 4, 6, 8, 10, 12, 14)]:
        with Context(**kw) as c:
            bt, bb = None, None
            for _ in range(5):
                t0 = time.perf_counter(); c.tree(build, tar); dt = time.perf_counter() - t0
                if bt is None or dt < bt[0]: bt = (dt, c.stats_ex())
            for _ in range(3):
                t0 = time.perf_counter(); assert _lib.lib().snaphash_sha512_buffers(c._h, ptrs, lens, n + 1, out) == 0; dt = time.perf_counter() - t0
                if bb is None or dt < bb[0]: bb = (dt, c.stats_ex())
        print("%-16s tree %.1f ms = %.1f GiB/s (host: %d streams, busiest thread %.0f ms; planned gpu %.0f host %.0f, took gpu %.0f, %d threads)   buffers %.1f ms = %.1f GiB/s (host: %d streams, %.0f ms)" %
              (label, bt[0] * 1e3, total / 2**30 / bt[0], bt[1]["host_streams"], bt[1]["host_ms"], bt[1]["planned_gpu_ms"], bt[1]["planned_host_ms"], bt[1]["gpu_ms"], bt[1]["host_threads_run"], bb[0] * 1e3, total / 2**30 / bb[0], bb[1]["host_streams"], bb[1]["host_ms"]), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
