#!/usr/bin/env python3
"""The DEFLATE kernel on three kinds of content, 64 MiB each: kernel time, output ratio, and zlib -6 / -9 on the same
bytes (first 8 MiB: zlib is slow) for comparison.  Zipf-word text (synthetic), this repo's sources (a tar of the tracked
text files, repeated with a running counter so that repeats lie beyond the window), binaries (libsnaphash.so + the
python interpreter, likewise).  usage: tools/deflate_corpora.py"""
import io
import os
import subprocess
import sys
import tarfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context  # noqa: E402

SIZE = 64 << 20


def text():
    rng = np.random.default_rng(5)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    return b" ".join(words[int(i)] for i in rng.zipf(1.3, size=SIZE // 5 + 16) % 2000)[:SIZE]


def sources():
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w") as tf:
        for d, _, files in sorted(os.walk(ROOT)):
            if any(part in d for part in (".git", "gpurun_out", "__pycache__", "variants", "profiles")):
                continue
            for f in sorted(files):
                if f.endswith((".py", ".cpp", ".h", ".hip", ".inc", ".c", ".md", ".sh")) or f == "Makefile":
                    tf.add(os.path.join(d, f), arcname=os.path.relpath(os.path.join(d, f), ROOT))
    return buf.getvalue()


def binaries():
    out = open(os.path.join(ROOT, "snappy_amd", "libsnaphash.so"), "rb").read()
    exe = os.path.realpath(sys.executable)
    return out + open(exe, "rb").read()


def fill(unit):
    # distinct "files" of the same kind: every repeat of the unit is salted so that it does not match the one 64 KiB before it
    parts, n, k = [], 0, 0
    rng = np.random.default_rng(11)
    while n < SIZE:
        salt = rng.integers(0, 256, size=max(64, len(unit) // 200), dtype=np.uint8).tobytes()
        parts.append(unit)
        parts.append(salt)
        n += len(unit) + len(salt)
        k += 1
    return b"".join(parts)[:SIZE]


depths = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
corpora = (("Zipf-word text", text()), ("this repo's sources (tar)", fill(sources())), ("binaries (.so + python)", fill(binaries())))
for depth in depths:
  if len(depths) > 1:
      print("## deflate_depth = %d" % (depth or 32), flush=True)
  with Context(deflate_depth=depth) as c:
    for name, data in corpora:
        c.gzip_buffer(data[:1 << 20])
        best = None
        for _ in range(3):
            gz = c.gzip_buffer(data)
            ms = c.targz_stats()["deflate_ms"]
            best = ms if best is None else min(best, ms)
        assert zlib.decompressobj(-15).decompress(gz[10:-8]) == data
        sample = data[:8 << 20]
        t0 = time.perf_counter()
        z6 = len(zlib.compress(sample, 6)) / len(sample)
        t6 = time.perf_counter() - t0
        z9 = len(zlib.compress(sample, 9)) / len(sample)
        gs = len(c.gzip_buffer(sample)) / len(sample)
        import hashlib
        print("%-28s kernel %.1f ms per 64 MiB = %.2f GB/s; ratio %.4f (first 8 MiB: %.4f; zlib -6 %.4f at %.0f MB/s on one core, zlib -9 %.4f); sha256 of the output %s" %
              (name, best, len(data) / best / 1e6, len(gz) / len(data), gs, z6, len(sample) / t6 / 1e6, z9, hashlib.sha256(gz).hexdigest()[:16]), flush=True)
