#!/usr/bin/env python3
"""Row f3 measured: the fused Build pass (tar + GPU DEFLATE + archive digest + per-file SHA-512 + hashes.yaml,
one read) on on-disk trees in tmpfs, next to what the reference does with the same tree on one host core
(tar stream through zlib level 9 -- compress/gzip level 9's algorithm -- then two SHA-512 passes; timed on a
bounded sample).  usage: tools/targz_bench.py [text|binary|mixed] [MiB total] [MiB per file]"""
import gzip
import hashlib
import io
import json
import os
import shutil
import sys
import tarfile
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context, synthetic  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
total_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
file_mib = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0


def text_block(rng, n):
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    idx = rng.zipf(1.3, size=n // 5 + 16) % 2000
    return b" ".join(words[int(i)] for i in idx)[:n]


base = "/dev/shm" if os.path.isdir("/dev/shm") else None
tmp = tempfile.mkdtemp(prefix="snaphash_f3_", dir=base)
try:
    build = os.path.join(tmp, "build")
    os.makedirs(os.path.join(build, "DEBIAN"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("Package: bench\n")
    rng = np.random.default_rng(5)
    fsize = int(file_mib * (1 << 20))
    nfiles = max(1, (total_mib << 20) // fsize)
    tblock = text_block(rng, min(fsize, 4 << 20))
    t0 = time.perf_counter()
    for i in range(nfiles):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        is_text = kind == "text" or (kind == "mixed" and i % 2 == 0)
        if is_text:
            off = int(rng.integers(0, max(1, len(tblock) - 1)))
            data = (tblock[off:] + tblock * (fsize // len(tblock) + 1))[:fsize]
        else:
            data = synthetic.file_bytes(fsize, i)
        with open(p, "wb") as f:
            f.write(data)
    print("tree: %d files x %d B (%s) in %.1f s" % (nfiles, fsize, kind, time.perf_counter() - t0), flush=True)
    out = os.path.join(tmp, "data.tar.gz")
    with Context() as c:
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            dt = time.perf_counter() - t0
            st, zs, ex = c.stats(), c.targz_stats(), c.stats_ex()
            print("fused pass %d: %.3f s = %.2f GiB/s of tree  (fill %.0f ms, h2d %.0f ms, deflate kernels %.0f ms, sha kernels %.0f ms, "
                  "gz %.1f MiB = ratio %.3f, stored chunks %d/%d)" % (
                      rep, dt, zs["tar_bytes"] / 2**30 / dt, zs["fill_ms"], st["h2d_ms"], zs["deflate_ms"], st["kernel_ms"],
                      zs["gz_bytes"] / 2**20, zs["gz_bytes"] / zs["tar_bytes"], zs["stored_chunks"], zs["chunks"]), flush=True)
            if best is None or dt < best[0]:
                best = (dt, st, zs)
        # GPU-only rate of the compressor: bytes in / kernel time
        dt, st, zs = best
        print("deflate kernels alone: %.1f GB/s of input (%.0f ms for %.1f MiB)" % (
            zs["tar_bytes"] / (zs["deflate_ms"] * 1e-3) / 1e9, zs["deflate_ms"], zs["tar_bytes"] / 2**20), flush=True)
        assert hashlib.sha512(open(out, "rb").read()).digest() == dig
    # the unfused alternative: the same tree and the finished archive hashed by a separate pass (hybrid scheduling on,
    # so that the one long archive stream finishes on a host thread instead of one GPU lane pair)
    with Context(host_threads=16) as c:
        t0 = time.perf_counter()
        y2 = c.tree(build, out)
        print("hash pass alone afterwards (second read of every file, host_threads=16): %.3f s; yaml identical: %s" % (
            time.perf_counter() - t0, y2 == y))
        assert y2 == y
    # validity: gzip + tarfile read it back (first members only, to bound the time)
    tf = tarfile.open(out, "r:gz")
    for k, m in enumerate(tf):
        if k >= 40:
            break
        if m.isreg():
            assert tf.extractfile(m).read() == open(os.path.join(build, m.name[2:]), "rb").read()
    # CPU baseline on a bounded sample: the reference's serial Build = tar + gzip -9, then SHA-512 of the archive and of every file
    sample_files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(build) if "DEBIAN" not in dp for f in fs)[:max(2, (64 << 20) // fsize)]
    blob = b"".join(open(p, "rb").read() for p in sample_files)
    t0 = time.perf_counter()
    z = zlib.compress(blob, 9)
    t_z = time.perf_counter() - t0
    t0 = time.perf_counter()
    hashlib.sha512(z).digest()
    for p in sample_files:
        hashlib.sha512(open(p, "rb").read()).digest()
    t_h = time.perf_counter() - t0
    print("CPU, 1 core, sample of %d files / %.0f MiB: zlib level 9 %.2f s (%.1f MB/s, ratio %.3f) + SHA-512 passes %.2f s -> %.3f GiB/s of tree" % (
        len(sample_files), len(blob) / 2**20, t_z, len(blob) / t_z / 1e6, len(z) / len(blob), t_h, len(blob) / 2**30 / (t_z + t_h)))
    print(json.dumps({"kind": kind, "tree_bytes": int(best[2]["tar_bytes"]), "fused_s": round(best[0], 4),
                      "fused_GiBps": round(best[2]["tar_bytes"] / 2**30 / best[0], 3), "deflate_ms": round(best[2]["deflate_ms"], 2),
                      "deflate_GBps_in": round(best[2]["tar_bytes"] / (best[2]["deflate_ms"] * 1e-3) / 1e9, 2),
                      "ratio": round(best[2]["gz_bytes"] / best[2]["tar_bytes"], 4),
                      "cpu_1core_GiBps": round(len(blob) / 2**30 / (t_z + t_h), 4), "cpu_zlib9_ratio": round(len(z) / len(blob), 4)}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
