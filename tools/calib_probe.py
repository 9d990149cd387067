#!/usr/bin/env python3
"""What a ctx measures about its box (planner.h PlanCalib) and how the plan's prediction compares with the call:
init probe, then n x 1 MiB from memory and from tmpfs files in the default configuration, several passes each.
usage: tools/calib_probe.py [n=4096]"""
import ctypes, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from snappy_amd import Context, _lib, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob = np.random.default_rng(6).integers(0, 256, size=(n << 20) + 4096, dtype=np.uint8)
ptrs = (ctypes.c_void_p * n)(*[blob.ctypes.data + (i << 20) + i % 4096 for i in range(n)])
lens = (ctypes.c_uint64 * n)(*[1 << 20] * n)
out = ctypes.create_string_buffer(64 * n)
tmp = tempfile.mkdtemp(prefix="snaphash_cp_", dir="/dev/shm")
try:
    build = os.path.join(tmp, "build")
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        blob[(i << 20):(i << 20) + (1 << 20)].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    blob[:1 << 20].tofile(tar)
    for threads in (0, 4):
        with Context(flags=0, host_threads=threads) as c:
            print("host_threads=%d: init calib %s" % (threads, {k: (round(v / 1e9, 2) if isinstance(v, float) else v) for k, v in c.calib().items()}))
            for what in ("memory", "files"):
                for rep in range(4):
                    t0 = time.perf_counter()
                    if what == "memory":
                        assert _lib.lib().snaphash_sha512_buffers(c._h, ptrs, lens, n, out) == 0
                    else:
                        c.tree(build, tar)
                    dt = (time.perf_counter() - t0) * 1e3
                    ex = c.stats_ex()
                    k = c.calib()
                    print("  %s pass %d: %.1f ms; planned gpu %.1f host %.1f | actual gpu %.1f host %.1f hash %.1f | host streams %d of %d, threads %d | calib dma %.1f fill_mem %.2f fill_files %.2f GB/s host_gain %.2f" %
                          (what, rep, dt, ex["planned_gpu_ms"], ex["planned_host_ms"], ex["gpu_ms"], ex["host_ms"], ex["hash_ms"], ex["host_streams"], n + (what == "files"),
                           ex["host_threads_run"], k["dma"] / 1e9, k["fill_mem"] / 1e9, k["fill_files"] / 1e9, k["host_gain"]), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
