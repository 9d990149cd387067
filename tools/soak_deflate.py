#!/usr/bin/env python3
"""Soak of the DEFLATE kernel: randomly structured inputs of random sizes (copies from every distance up to beyond the
window, runs, literal bursts, few-symbol text, mutating phrases; from a few bytes to several MiB, across staging
pieces of random size) through snaphash_gzip_buffer; every output must inflate to its input (zlib) AND equal the
serial CPU model byte for byte.  The workgroup's pipeline hands tiles out from queues and overlaps four stages: a
race would show here as a rare mismatch.  usage: tools/soak_deflate.py [seconds]"""
import ctypes
import os
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
so = "/tmp/f3_harness_soak.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "f3_host_harness.cpp"), "-pthread"])
f3 = ctypes.CDLL(so)
f3.f3_model_gzip2.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
f3.f3_model_gzip2.restype = ctypes.c_void_p
f3.f3_model_gzip3.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t)]
f3.f3_model_gzip3.restype = ctypes.c_void_p
f3.f3_free.argtypes = [ctypes.c_void_p]
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))


def make(target):
    buf = bytearray()
    while len(buf) < target:
        kind = int(rng.integers(0, 6))
        if kind == 0 or len(buf) < 8:
            buf += rng.integers(0, 256, size=int(rng.integers(1, 3000)), dtype=np.uint8).tobytes()
        elif kind == 1:
            buf += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 5000))
        elif kind == 2:
            dist = int(rng.integers(1, min(len(buf), 70000) + 1))
            n = int(rng.integers(3, 3000))
            for _ in range(n):
                buf.append(buf[-dist])
        elif kind == 3:
            buf += bytes(rng.choice(np.frombuffer(b"etaoin shrdlu\n", dtype=np.uint8), size=int(rng.integers(1, 20000))))
        elif kind == 4:
            phrase = bytearray(rng.integers(97, 123, size=int(rng.integers(4, 60)), dtype=np.uint8).tobytes())
            for _ in range(int(rng.integers(1, 300))):
                phrase[int(rng.integers(0, len(phrase)))] = int(rng.integers(97, 123))
                buf += phrase
        else:  # a block of earlier content again, far back
            if len(buf) > 100:
                a = int(rng.integers(0, len(buf) - 50))
                buf += buf[a:a + int(rng.integers(50, 40000))]
    return bytes(buf[:target])


t_end = time.time() + budget
t_note = time.time() + 60.0
it = 0
replay = int(os.environ.get("SOAK_REPLAY", "-1"))  # regenerate the inputs up to this iteration without compressing them, then hammer that one
total = 0
ctxs = {}
while time.time() < t_end:
    staging = int(rng.choice([1 << 16, 1 << 17, 3 << 16, 1 << 20, 1 << 22]))
    target = int(rng.choice([1, 2, 3, 63, 64, 65, 1919, 1920, 1921, 65535, 65536, 65537, int(rng.integers(1, 300000)), int(rng.integers(1, 6 << 20))]))
    data = make(target)
    if replay >= 0 and it < replay:
        it += 1
        continue
    depth = int(rng.choice([0, 0, 0, 4, 8, 48, 64, 128]))  # snaphash_config.deflate_depth (0 = the default, 32)
    c = ctxs.get((staging, depth)) or ctxs.setdefault((staging, depth), Context(staging_bytes=staging, deflate_depth=depth))
    if replay >= 0:
        n = ctypes.c_size_t()
        p = f3.f3_model_gzip3(data, len(data), staging, depth, ctypes.byref(n))
        model = ctypes.string_at(p, n.value)
        f3.f3_free(p)
        outs = [c.gzip_buffer(data) for _ in range(300)]
        bad = [k for k, g in enumerate(outs) if g != model]
        print("replay of iteration %d: %d bytes, staging %d: %d of 300 runs differ from the model (%s); distinct outputs %d" %
              (it, len(data), staging, len(bad), bad[:10], len(set(outs))), flush=True)
        open(os.path.join(ROOT, "gpurun_out", "soak_replay_input.bin"), "wb").write(data)
        if bad:
            g = outs[bad[0]]
            first = next(i for i in range(min(len(g), len(model))) if g[i] != model[i])
            print("  first differing byte %d of %d / %d; inflates: %s" % (first, len(g), len(model), zlib.decompressobj(-15).decompress(g[10:-8]) == data))
        break
    gz = c.gzip_buffer(data)
    assert zlib.decompressobj(-15).decompress(gz[10:-8]) == data, ("inflate", it, target, staging, depth)
    n = ctypes.c_size_t()
    p = f3.f3_model_gzip3(data, len(data), staging, depth, ctypes.byref(n))
    model = ctypes.string_at(p, n.value)
    f3.f3_free(p)
    assert gz == model, ("model", it, target, staging, depth)
    it += 1
    total += len(data)
    if time.time() >= t_note:  # (a run that says nothing for minutes is taken for hung)
        print("  ... %d inputs, %.1f MiB so far" % (it, total / 2**20), flush=True)
        t_note = time.time() + 60.0
for c in ctxs.values():
    c.close()
print("soak_deflate: %d inputs, %.1f MiB (search depths 4 / 8 / 32 / 48 / 64 / 128 at random), all inflate to their input and equal the CPU model" % (it, total / 2**20))
