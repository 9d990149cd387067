#!/usr/bin/env python3
"""BASELINE config 5 (100 000 Zipf files, 3.0 GiB) as a real on-disk tree (tmpfs) -> hashes.yaml, GPU only and as
planned (ABI 4): where does the time go once the 256 MiB head file no longer sets it?  usage: tools/c5_tree_hybrid.py"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from snappy_amd import Context, synthetic  # noqa: E402

sizes = synthetic.config_sizes("C5")
off, total = synthetic.pack_offsets(sizes)
with Context(device=0) as c0:
    dev = torch.empty(total, dtype=torch.uint8, device="cuda")
    c0.fill_synthetic_device(dev.data_ptr(), off, sizes, np.arange(len(sizes), dtype=np.uint64))
    host = dev.cpu().numpy()
del dev
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
tmp = tempfile.mkdtemp(prefix="snaphash_c5_", dir=base)
try:
    build = os.path.join(tmp, "build")
    n = len(sizes) - 1
    t0 = time.perf_counter()
    made = set()
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        d = os.path.dirname(p)
        if d not in made:
            os.makedirs(d, exist_ok=True)
            made.add(d)
        host[int(off[i]):int(off[i]) + int(sizes[i])].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    host[int(off[n]):int(off[n]) + int(sizes[n])].tofile(tar)
    print("tree of %d files, %.2f GiB, written in %.1f s" % (n + 1, total / 2**30, time.perf_counter() - t0), flush=True)
    ref = None
    from snappy_amd import _lib
    for name, kw, reps in (("GPU only", dict(flags=_lib.FLAG_GPU_ONLY), 1), ("default (planned)", dict(flags=0), 5), ("host_threads=4", dict(flags=0, host_threads=4), 3)) + tuple(
            ("host_threads=%d" % int(a), dict(flags=0, host_threads=int(a)), 3) for a in sys.argv[1:]):
        with Context(**kw) as c:
            best = None
            for rep in range(reps):
                t0 = time.perf_counter()
                y = c.tree(build, tar)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, c.stats(), c.stats_ex())
            dt, st, ex = best
            ref = ref or y
            print("%-18s %.3f s = %.2f GiB/s  (kernels %.0f ms, h2d %.0f ms, host streams %d / %.0f MiB, yaml %s)" % (
                name + ":", dt, total / 2**30 / dt, st["kernel_ms"], st["h2d_ms"], ex["host_streams"], ex["host_bytes"] / 2**20,
                "identical" if y == ref else "DIFFERS"), flush=True)
            t0 = time.perf_counter()
            assert c.verify(build, y, tar) is None
            print("                   verify %.3f s" % (time.perf_counter() - t0), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
