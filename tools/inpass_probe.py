import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from snappy_amd import Context, synthetic
print(bench.bind_to_gpu_node(0))
tmp = tempfile.mkdtemp(prefix="snaphash_inpass_", dir="/dev/shm")
try:
    build = os.path.join(tmp, "build"); os.makedirs(os.path.join(build, "DEBIAN"))
    rng = np.random.default_rng(5)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    block = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=(4 << 20) // 5 + 16) % 2000)[:4 << 20]
    for i in range(1024):
        p = os.path.join(build, synthetic.file_name(i)); os.makedirs(os.path.dirname(p), exist_ok=True)
        off = int(rng.integers(0, len(block) - 1))
        open(p, "wb").write((block[off:] + block)[:1 << 20])
    out = os.path.join(tmp, "data.tar.gz")
    whole = b"".join(open(os.path.join(build, synthetic.file_name(i)), "rb").read() for i in range(1024))
    with Context() as c:
        for with_hashes in (True, False, True, False):
            best = None
            for _ in range(3):
                t0 = time.perf_counter(); c.tar_create(out, build, build + "/DEBIAN", with_hashes=with_hashes); dt = time.perf_counter() - t0
                zs = c.targz_stats(); st = c.stats()
                if best is None or dt < best[0]: best = (dt, zs["deflate_ms"], st["kernel_ms"], zs["fill_ms"])
            print("tar_create with_hashes=%s: %.1f ms, DEFLATE kernels %.1f ms, SHA-512 kernels %.1f ms, fill %.1f ms" % ((with_hashes,) + tuple(x * (1e3 if k == 0 else 1) for k, x in enumerate(best))), flush=True)
        for _ in range(2):
            t0 = time.perf_counter(); gz = c.gzip_buffer(whole); dt = time.perf_counter() - t0
            zs = c.targz_stats()
            print("gzip_buffer of the same GiB from memory: %.1f ms, DEFLATE kernels %.1f ms" % (dt * 1e3, zs["deflate_ms"]), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
