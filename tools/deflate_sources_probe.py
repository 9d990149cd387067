#!/usr/bin/env python3
"""64 MiB of this repository's sources (the corpus of tools/deflate_corpora.py) through snaphash_gzip_buffer a few times: the
workload for rocprofv3 counter passes on deflate_chunks_kernel (SNAPHASH_LIB picks the build: make narrow for A/B)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import io, tarfile
import numpy as np
from snappy_amd import Context

SIZE = 64 << 20
buf = io.BytesIO()
with tarfile.open(fileobj=buf, mode="w") as tf:
    for d, _, files in sorted(os.walk(ROOT)):
        if any(part in d for part in (".git", "gpurun_out", "__pycache__", "variants", "profiles")):
            continue
        for f in sorted(files):
            if f.endswith((".py", ".cpp", ".h", ".hip", ".inc", ".c", ".md", ".sh")) or f == "Makefile":
                tf.add(os.path.join(d, f), arcname=os.path.relpath(os.path.join(d, f), ROOT))
unit = buf.getvalue()
parts, n = [], 0
rng = np.random.default_rng(11)
while n < SIZE:
    salt = rng.integers(0, 256, size=max(64, len(unit) // 200), dtype=np.uint8).tobytes()
    parts += [unit, salt]
    n += len(unit) + len(salt)
data = b"".join(parts)[:SIZE]
with Context() as c:
    for _ in range(4):
        gz = c.gzip_buffer(data)
        print("deflate kernels %.2f ms, ratio %.4f" % (c.targz_stats()["deflate_ms"], len(gz) / len(data)), flush=True)
