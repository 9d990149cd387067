#!/usr/bin/env python3
"""Soak of the data.tar.gz producer at its default staging size: random trees of 70-400 MiB (a few big members that straddle
the 64 MiB parts of the first slot and the 256 MiB slots, many small ones, empty files, long names), every pass checked:
the archive inflates to the tar stream the host model lays out, archive digest = hashlib, hashes.yaml = the oracle's.
usage: tools/soak_tar.py [seconds=240] [seed=1] [mixed]
"mixed" (round 5): small packages (0.05-40 MiB, what most snaps are) between the large ones, in ONE ctx: the producer's buffers are
sized for the job and grow when a later one needs more, and long members are hashed by host threads out of the staging slots."""
import ctypes, gzip, hashlib, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context  # noqa: E402
from oracle import oracle  # noqa: E402  (the checker)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mixed = len(sys.argv) > 3 and sys.argv[3] == "mixed"
tmp = tempfile.mkdtemp(prefix="snaphash_soaktar_", dir="/dev/shm")
sodir = tempfile.mkdtemp(prefix="snaphash_soaktar_so_")  # (/dev/shm is mounted noexec on the GPU box)
so = os.path.join(sodir, "libf3host.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "f3_host_harness.cpp")])
L = ctypes.CDLL(so)
L.f3_tar_stream.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
L.f3_free.argtypes = [ctypes.c_void_p]
MiB = 1 << 20
pool = rng.integers(0, 256, size=64 * MiB, dtype=np.uint8)
words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(400)]
text = np.frombuffer(b" ".join(words[int(i)] for i in rng.zipf(1.3, size=2000000) % 400), dtype=np.uint8)
t_end, it, total = time.time() + budget, 0, 0
try:
    with Context() as c:
        while time.time() < t_end:
            build = os.path.join(tmp, "t%d" % it)
            os.makedirs(os.path.join(build, "DEBIAN"))
            open(os.path.join(build, "DEBIAN", "control"), "w").write("Package: soak\n")
            small = mixed and rng.random() < 0.7
            want = int(rng.integers(50 << 10, 40 * MiB)) if small else int(rng.integers(70, 400)) * MiB
            have, k = 0, 0
            while have < want:
                kind = rng.random()
                if small:
                    size = int(rng.integers(1, 12) * MiB + rng.integers(0, 4096)) if kind < 0.05 else int(rng.choice([0, 1, 511, 512, 513, 65535, 65536])) if kind < 0.3 else int(rng.lognormal(np.log(20000), 1.5))
                else:
                    size = int(rng.integers(20, 130) * MiB + rng.integers(0, 4096)) if kind < 0.15 else int(rng.choice([0, 1, 511, 512, 513, 65535, 65536])) if kind < 0.3 else int(rng.integers(0, 3 * MiB))
                src = text if rng.random() < 0.5 else pool
                off = int(rng.integers(0, max(1, len(src) - min(size, len(src)))))
                d = os.path.join(build, "d%d" % int(rng.integers(0, 5)))
                os.makedirs(d, exist_ok=True)
                name = "f%04d" % k if rng.random() < 0.9 else "n" * int(rng.integers(101, 200)) + "%d" % k
                with open(os.path.join(d, name), "wb") as f:
                    left = size
                    while left > 0:
                        n = min(left, len(src) - off)
                        f.write(src[off:off + n].tobytes())
                        left -= n
                        off = 0
                have += size
                k += 1
            p, n = ctypes.c_void_p(), ctypes.c_size_t()
            assert L.f3_tar_stream(build.encode(), (build + "/DEBIAN").encode(), ctypes.byref(p), ctypes.byref(n)) == 0
            want_tar = ctypes.string_at(p.value, n.value)
            L.f3_free(p)
            out = os.path.join(tmp, "o.tar.gz")
            y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            raw = open(out, "rb").read()
            assert gzip.decompress(raw) == want_tar, ("tar stream differs", it)
            assert hashlib.sha512(raw).digest() == dig, ("archive digest", it)
            assert y == oracle.hashes_yaml(build, out), ("hashes.yaml", it)
            total += len(want_tar)
            it += 1
            shutil.rmtree(build)
            if it % 5 == 0:
                print("  ... %d trees, %.1f GiB of tar stream so far" % (it, total / 2**30), flush=True)
    print("soak_tar: %d trees, %.1f GiB of tar stream (%s, default staging: first slot in 64 MiB parts), every archive inflates to its tar stream, "
          "digest = hashlib, hashes.yaml = the oracle's" % (it, total / 2**30, "0.05-40 MiB and 70-400 MiB mixed in one ctx" if mixed else "70-400 MiB each"))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(sodir, ignore_errors=True)
