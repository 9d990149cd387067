#!/usr/bin/env python3
"""Soak of the host-side engine (sub-slot batches, tapered ends, descriptor cache, the planner): random stream lists of
random shapes through snaphash_sha512_buffers / _files with random staging sizes, engines and configurations; every
digest against hashlib.  usage: tools/soak_engine.py [seconds=240] [seed=1]"""
import ctypes, hashlib, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from snappy_amd import Context, _lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pool = rng.integers(0, 256, size=(96 << 20) + 4096, dtype=np.uint8)
L = _lib.lib()
t_end = time.time() + budget
it, total_bytes, total_streams, kinds = 0, 0, 0, {}
tmp = tempfile.mkdtemp(prefix="snaphash_soak_", dir="/dev/shm")
try:
    while time.time() < t_end:
        shape = rng.choice(["equal", "zipf", "tiny", "few-huge", "mixed"])
        n = int(rng.choice([1, 2, 7, 64, 65, 300, 1250, 2049, 4097, 6000]))
        if shape == "equal":
            sizes = np.full(n, int(rng.choice([0, 1, 127, 128, 129, 4096, 65536, 1 << 20])), dtype=np.int64)
        elif shape == "zipf":
            sizes = np.minimum((1 << 24) // np.arange(1, n + 1), 1 << 24).astype(np.int64) - rng.integers(0, 100, size=n) % 97
            sizes = np.maximum(sizes, 0)
        elif shape == "tiny":
            sizes = rng.integers(0, 300, size=n)
        elif shape == "few-huge":
            n = int(rng.integers(1, 6)); sizes = rng.integers(1 << 20, 40 << 20, size=n)
        else:
            sizes = np.concatenate([rng.integers(0, 5000, size=n), rng.integers(1 << 16, 3 << 20, size=max(1, n // 50))])
        while sizes.sum() > (700 << 20):
            sizes = sizes[:max(1, len(sizes) // 2)]
        n = len(sizes)
        offs = rng.integers(0, 4096, size=n)
        bufs = [pool[int(o):int(o) + int(s)] if s <= (96 << 20) else None for o, s in zip(offs, sizes)]
        staging = int(rng.choice([1 << 16, 1 << 20, 8 << 20, 32 << 20, 64 << 20, 256 << 20]))
        flags = int(rng.choice([_lib.FLAG_GPU_ONLY, _lib.FLAG_GPU_ONLY, 0]))
        devices = [0, 0] if rng.random() < 0.25 else None
        files = rng.random() < 0.4 and n <= 6200  # (round 5: file sources of more than 2 048 streams -- the cap on files begun per batch, the cut of the last batch)
        kind = "%s/%s/%s/%s" % (shape, "files" if files else "mem", "gpu" if flags else "planned", "2eng" if devices else "1eng")
        kinds[kind] = kinds.get(kind, 0) + 1
        with Context(staging_bytes=staging, flags=flags, devices=devices) as c:
            if files:
                d = os.path.join(tmp, "i%d" % it)
                os.makedirs(d)
                paths = []
                for k, b in enumerate(bufs):
                    p = os.path.join(d, "f%05d" % k)
                    b.tofile(p)
                    paths.append(p)
                got = c.sha512_files(paths)
                shutil.rmtree(d)
            else:
                ptrs = (ctypes.c_void_p * n)(*[pool.ctypes.data + int(o) for o in offs])
                clens = (ctypes.c_uint64 * n)(*[int(s) for s in sizes])
                out = ctypes.create_string_buffer(64 * n)
                rc = L.snaphash_sha512_buffers(c._h, ptrs, clens, n, out)
                assert rc == 0, rc
                got = [out.raw[64 * k:64 * k + 64] for k in range(n)]
            ex = c.stats_ex()
            assert ex["gpu_bytes"] + ex["host_bytes"] == int(sizes.sum()), (ex, int(sizes.sum()))
            if flags:
                assert ex["host_bytes"] == 0
        for k in range(n):
            if got[k] != hashlib.sha512(bufs[k]).digest():
                raise SystemExit("MISMATCH: iteration %d (%s), stream %d of %d, size %d, staging %d" % (it, kind, k, n, sizes[k], staging))
        it += 1
        total_bytes += int(sizes.sum())
        total_streams += n
        if it % 25 == 0:
            print("%d iterations, %d streams, %.1f GiB, all digests = hashlib" % (it, total_streams, total_bytes / 2**30), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
print("soak ok: %d iterations, %d streams, %.1f GiB; combinations seen: %d" % (it, total_streams, total_bytes / 2**30, len(kinds)))
for k in sorted(kinds):
    print("  %-40s %d" % (k, kinds[k]))
