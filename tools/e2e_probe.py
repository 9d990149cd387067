#!/usr/bin/env python3
"""End-to-end tuning probe (row a2 on an on-disk config-2 tree, and host buffers): one tree in /dev/shm, then one child
process per setting (SNAPHASH_COPY_THREADS is read once per process) that runs snaphash_tree / snaphash_sha512_buffers
three times with SNAPHASH_TRACE_TREE=1.  usage: tools/e2e_probe.py [settings "files:12,files:24,mem:6,..."]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "child":
    import ctypes
    import numpy as np
    from snappy_amd import Context, _lib, synthetic
    kind, build, tar = sys.argv[2], sys.argv[3], sys.argv[4]
    lens = synthetic.config_sizes("C2")
    total = int(lens.sum())
    with Context(flags=_lib.FLAG_GPU_ONLY) as c:
        if kind == "files":
            for rep in range(3):
                t0 = time.perf_counter()
                y = c.tree(build, tar)
                dt = time.perf_counter() - t0
                st = c.stats()
                print("  tree pass %d: %.1f ms = %.2f GiB/s (kernel %.1f ms, h2d %.1f ms, %d launches)" %
                      (rep, dt * 1e3, total / 2**30 / dt, st["kernel_ms"], st["h2d_ms"], st["launches"]), flush=True)
        else:
            off, tot = synthetic.pack_offsets(lens)
            host = np.empty(tot, dtype=np.uint8)
            host[:] = 7
            n = len(lens)
            ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + int(o) for o in off])
            clens = (ctypes.c_uint64 * n)(*[int(x) for x in lens])
            out = ctypes.create_string_buffer(64 * n)
            for rep in range(3):
                t0 = time.perf_counter()
                rc = _lib.lib().snaphash_sha512_buffers(c._h, ptrs, clens, n, out)
                dt = time.perf_counter() - t0
                st = c.stats()
                print("  buffers pass %d: %.1f ms = %.2f GiB/s (kernel %.1f ms, h2d %.1f ms, %d launches) rc %d" %
                      (rep, dt * 1e3, total / 2**30 / dt, st["kernel_ms"], st["h2d_ms"], st["launches"], rc), flush=True)
    sys.exit(0)

import numpy as np
import torch
from snappy_amd import Context, synthetic
settings = (sys.argv[1] if len(sys.argv) > 1 else "files:12,files:16,files:24,mem:6,mem:8,mem:12").split(",")
lens = synthetic.config_sizes("C2")
off, total = synthetic.pack_offsets(lens)
tmp = tempfile.mkdtemp(prefix="snaphash_probe_", dir="/dev/shm")
try:
    with Context(device=0) as c0:
        dev = torch.empty(total, dtype=torch.uint8, device="cuda")
        c0.fill_synthetic_device(dev.data_ptr(), off, lens, np.arange(len(lens), dtype=np.uint64))
        host = dev.cpu().numpy()
        del dev
    build = os.path.join(tmp, "build")
    made = set()
    for i in range(len(lens) - 1):
        p = os.path.join(build, synthetic.file_name(i))
        d = os.path.dirname(p)
        if d not in made:
            os.makedirs(d, exist_ok=True)
            made.add(d)
        host[int(off[i]):int(off[i]) + int(lens[i])].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    host[int(off[-1]):int(off[-1]) + int(lens[-1])].tofile(tar)
    del host
    torch.cuda.empty_cache()
    for s in settings:
        kind, thr = s.split(":")[0], s.split(":")[1]
        extra = s.split(":")[2:]  # further KEY=VALUE environment settings
        env = dict(os.environ, SNAPHASH_COPY_THREADS=thr, SNAPHASH_TRACE_TREE="1")
        for kv in extra:
            k, v = kv.split("=")
            env[k] = v
        print("== %s" % s, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", kind, build, tar], env=env, check=False)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
