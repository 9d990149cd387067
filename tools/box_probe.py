#!/usr/bin/env python3
"""What the GPU box gives a process, and what a small call costs on it: the inputs of the planner's cost model
(plan_host_streams, snaphash_api.cpp).  usage: tools/box_probe.py   (prints; keep as profiles/r04_box_probe.txt)"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cat(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return "<%s>" % e.strerror


print("nproc (affinity): %d of %d; cgroup cpu.max: %s; cpuset.cpus.effective: %s" %
      (len(os.sched_getaffinity(0)), os.cpu_count(), cat("/sys/fs/cgroup/cpu.max"), cat("/sys/fs/cgroup/cpuset.cpus.effective")))
print("memory.max: %s; MemTotal: %s; /dev/shm free: %.0f GiB" %
      (cat("/sys/fs/cgroup/memory.max"), cat("/proc/meminfo").splitlines()[0], __import__("shutil").disk_usage("/dev/shm").free / 2**30))
print("numa nodes:", [d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")], flush=True)
exe = os.path.join(ROOT, "tools", "hostsha_threads")
if os.path.exists(exe):
    print(subprocess.run([exe, "32"], stdout=subprocess.PIPE).stdout.decode(), flush=True)

from snappy_amd import Context, _lib  # noqa: E402
L = _lib.lib()


def buffers(c, sizes, reps=7):
    n = len(sizes)
    host = np.random.default_rng(1).integers(0, 256, size=int(sum(sizes)) + 64, dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + int(o) for o in offs])
    clens = (ctypes.c_uint64 * n)(*[int(s) for s in sizes])
    out = ctypes.create_string_buffer(64 * n)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = L.snaphash_sha512_buffers(c._h, ptrs, clens, n, out)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    return ts, c.stats(), c.stats_ex()


t0 = time.perf_counter()
c = Context(flags=_lib.FLAG_GPU_ONLY)
print("snaphash_init (GPU only) %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for name, sizes in (("1 x 128 B", [128]), ("64 x 128 B", [128] * 64), ("1 x 256 KiB", [1 << 18]), ("1 x 1 MiB", [1 << 20]),
                    ("24 files incl. 1 MiB", [1000, 50000, 200000, 3, 1 << 20, 4096] * 4), ("5000 x 8 KiB", [8192] * 5000),
                    ("200 files incl. 3 MiB", [3 << 20] + [40000] * 199), ("1024 x 1 MiB", [1 << 20] * 1024)):
    ts, st, ex = buffers(c, sizes)
    print("GPU only  %-24s first %.2f ms, then min %.2f / median %.2f ms (kernel %.2f ms, h2d %.2f ms, %d launches)" %
          (name, ts[0], min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2], st["kernel_ms"], st["h2d_ms"], st["launches"]), flush=True)
c.close()
for ht in (1, 16):
    t0 = time.perf_counter()
    c = Context(flags=0, host_threads=ht)
    ti = (time.perf_counter() - t0) * 1e3
    for name, sizes in (("1 x 256 KiB", [1 << 18]), ("200 files incl. 3 MiB", [3 << 20] + [40000] * 199), ("64 x 64 MiB", [64 << 20] * 64)):
        ts, st, ex = buffers(c, sizes, reps=4)
        print("host_threads=%d init %.1f ms  %-24s first %.2f ms, then min %.2f ms; host bytes %d of %d, host_ms %.2f" %
              (ht, ti, name, ts[0], min(ts[1:]), ex["host_bytes"], st["bytes_hashed"], ex["host_ms"]), flush=True)
    c.close()
