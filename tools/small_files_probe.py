#!/usr/bin/env python3
"""Trees of many SMALL files (tmpfs) -> snaphash_tree in the default (planned) configuration: what the plan said the GPU and the
host part would take beside what they took, and what the ctx has calibrated afterwards.  usage: tools/small_files_probe.py [shape ...]
(SNAPHASH_TRACE_BATCHES=1 in the environment prints the engine's batches.)"""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.getcwd())
from snappy_amd import Context, _lib
tmp = tempfile.mkdtemp(prefix="snaphash_small_", dir="/dev/shm")
try:
    for shape, n, sz in (("5000x8KiB", 5000, 8192), ("20000x8KiB", 20000, 8192), ("5000x64KiB", 5000, 65536), ("50000x2KiB", 50000, 2048)):
        if len(sys.argv) > 1 and shape not in sys.argv[1:]:
            continue
        build = os.path.join(tmp, shape, "build")
        blob = np.random.default_rng(1).integers(0, 256, size=sz + 4096, dtype=np.uint8)
        for i in range(n):
            d = os.path.join(build, "d%04d" % (i // 100))
            if i % 100 == 0: os.makedirs(d)
            blob[i % 4096:(i % 4096) + sz].tofile(os.path.join(d, "f%06d.bin" % i))
        tar = os.path.join(tmp, shape, "data.tar.gz"); blob[:1000].tofile(tar)
        with Context() as c:
            c.tree(build, tar); c.tree(build, tar)
            rows = []
            for k in range(5):
                t0 = time.perf_counter(); c.tree(build, tar); dt = (time.perf_counter() - t0) * 1e3
                ex = c.stats_ex()
                rows.append((dt, ex["planned_gpu_ms"], ex["gpu_ms"], ex["planned_host_ms"], ex["host_ms"], ex["host_streams"], ex["planned_threads"]))
            rows.sort()
            print(shape, "tree %.2f ms; planned gpu %.2f took %.2f; planned host %.2f took %.2f; host streams %d threads %d" % rows[0], flush=True)
            print("   calib", c.calib(), flush=True)
        shutil.rmtree(os.path.join(tmp, shape))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
