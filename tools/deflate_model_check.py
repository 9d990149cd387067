#!/usr/bin/env python3
"""GPU DEFLATE output vs its CPU model (tests/f3_host_harness.cpp) on larger, repetitive inputs than the unit tests
use, twice (run-to-run determinism).  usage: tools/deflate_model_check.py [MiB]"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 24
so = os.path.join(tempfile.mkdtemp(), "libf3host.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "f3_host_harness.cpp")])
L = ctypes.CDLL(so)
L.f3_model_gzip2.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
L.f3_model_gzip2.restype = ctypes.c_void_p
L.f3_free.argtypes = [ctypes.c_void_p]
rng = np.random.default_rng(5)
words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
block = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=(4 << 20) // 5 + 16) % 2000)[:4 << 20]
pieces = []
while sum(map(len, pieces)) < (mib << 20):
    off = int(rng.integers(0, len(block) - 1))
    pieces.append((block[off:] + block)[:16384] + bytes(512))  # near-identical 16 KiB files with record padding between
data = b"".join(pieces)[:mib << 20]
inputs = {"repetitive 16 KiB files": data, "one block repeated": (block[:100000] * ((mib << 20) // 100000 + 1))[:mib << 20]}
staging = 4 << 20
with Context(staging_bytes=staging) as c:
    for name, d in inputs.items():
        a = c.gzip_buffer(d)
        b = c.gzip_buffer(d)
        n = ctypes.c_size_t()
        p = L.f3_model_gzip2(d, len(d), staging, ctypes.byref(n))
        m = ctypes.string_at(p, n.value)
        L.f3_free(p)
        first = next((i for i in range(min(len(a), len(m))) if a[i] != m[i]), None) if a != m else None
        print("%-26s %d B -> GPU %d B (ratio %.4f); second run identical: %s; equals the model: %s%s" % (
            name, len(d), len(a), len(a) / len(d), a == b, a == m, "" if a == m else " (model %d B, first difference at %s)" % (len(m), first)), flush=True)
        assert a == b and a == m
