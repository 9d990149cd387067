#!/usr/bin/env python3
"""How steady is the planned split of a link-bound tree on a 16-core quota?  The C2 tree (files on tmpfs) through
snaphash_tree, GPU only and in the default configuration, both contexts open (as in bench.py), alternating, every pass
printed.  usage: tools/default_variance.py [n=10000] [rounds=12]"""
import os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench  # noqa: E402
from snappy_amd import Context, _lib, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print(bench.bind_to_gpu_node(0))
tmp = tempfile.mkdtemp(prefix="snaphash_dv_", dir="/dev/shm")
try:
    build = os.path.join(tmp, "build")
    host = np.random.default_rng(3).integers(0, 256, size=(n + 1) << 20, dtype=np.uint8)
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        host[i << 20:(i + 1) << 20].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    host[n << 20:].tofile(tar)
    del host
    g = Context(flags=_lib.FLAG_GPU_ONLY)
    d = Context(flags=0)
    rows = {"GPU only": [], "default": []}
    for r in range(rounds):
        for name, c in (("GPU only", g), ("default", d)):
            t0 = time.perf_counter(); c.tree(build, tar); dt = time.perf_counter() - t0
            st, ex = c.stats(), c.stats_ex()
            rows[name].append(dt * 1e3)
            print("%-9s %.1f ms  (h2d busy %.0f ms, host part %.0f ms on %d streams)" % (name, dt * 1e3, st["h2d_ms"], ex["host_ms"], ex["host_streams"]), flush=True)
    for name, v in rows.items():
        v = sorted(v[1:])
        print("%-9s min %.1f  median %.1f  max %.1f ms over %d passes" % (name, v[0], v[len(v) // 2], v[-1], len(v)))
    g.close(); d.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
