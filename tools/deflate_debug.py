#!/usr/bin/env python3
"""GPU DEFLATE against its CPU model, sample by sample: sizes, first differing byte, whether zlib inflates either."""
import ctypes
import os
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from snappy_amd import Context  # noqa: E402

so = "/tmp/f3_harness.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "f3_host_harness.cpp"), "-pthread"])
f3 = ctypes.CDLL(so)
f3.f3_model_gzip2.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
f3.f3_model_gzip2.restype = ctypes.c_void_p
f3.f3_free.argtypes = [ctypes.c_void_p]

rng = np.random.default_rng(7)
words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(500)]
prose = b" ".join(words[int(i)] for i in rng.integers(0, 500, size=60000))
samples = {"one": b"x", "three": b"abc", "abcabc": b"abcabcabcabc", "64 x": b"x" * 64, "65 x": b"x" * 65, "zeros 1000": bytes(1000),
           "prose 100": prose[:100], "prose 4000": prose[:4000], "prose 4200": prose[:4200], "prose 9000": prose[:9000],
           "prose 65536": prose[:65536], "prose 70000": prose[:70000], "prose 200000": prose[:200000],
           "random 50000": rng.integers(0, 256, size=50000, dtype=np.uint8).tobytes()}
from test_f3_host import sample_inputs  # noqa: E402
samples = dict(sample_inputs())
samples["3 MiB mixed"] = (b"snappy " * 100000 + rng.integers(0, 256, size=1 << 20, dtype=np.uint8).tobytes() + bytes(1 << 20))[:3 << 20]
with Context(staging_bytes=1 << 20) as c:
    for name, data in samples.items():
        gz = c.gzip_buffer(data)
        n = ctypes.c_size_t()
        p = f3.f3_model_gzip2(data, len(data), 1 << 20, ctypes.byref(n))
        model = ctypes.string_at(p, n.value)
        f3.f3_free(p)
        def inflates(b):
            try:
                return zlib.decompressobj(-15).decompress(b[10:-8]) == data
            except zlib.error as e:
                return "zlib: %s" % e
        first = next((i for i in range(min(len(gz), len(model))) if gz[i] != model[i]), None)
        print("%-14s in %7d  gpu %7d  model %7d  equal %s  first diff %s  gpu inflates %s  model inflates %s" %
              (name, len(data), len(gz), len(model), gz == model, first, inflates(gz), inflates(model)), flush=True)
        if first is not None and len(data) < 300:
            print("   gpu  ", gz[10:].hex())
            print("   model", model[10:].hex())
