#!/usr/bin/env python3
"""Does the planner's choice hold?  Trees of several shapes (tmpfs) -> snaphash_tree in the DEFAULT configuration and with
SNAPHASH_FLAG_GPU_ONLY, best of 5 after two warm calls each, hashes.yaml compared; the default's prediction beside what its parts took.
A default that is slower than GPU-only by more than noise is a mis-plan.  usage: tools/choice_probe.py [shape ...]"""
import os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context, _lib  # noqa: E402

KiB, MiB = 1 << 10, 1 << 20
rng = np.random.default_rng(11)
SHAPES = {
    "snap-like 300 (lognormal, median 20 KiB)": np.minimum(64 * MiB, np.maximum(1, rng.lognormal(np.log(20 * KiB), 2.0, size=300))).astype(np.int64),
    "snap-like 3000 (lognormal, median 20 KiB)": np.minimum(64 * MiB, np.maximum(1, rng.lognormal(np.log(20 * KiB), 2.0, size=3000))).astype(np.int64),
    "2000 x 256 KiB": np.full(2000, 256 * KiB),
    "1000 x 1 MiB": np.full(1000, MiB),
    "3000 x 1 MiB": np.full(3000, MiB),
    "200 x 16 MiB": np.full(200, 16 * MiB),
    "10 x 100 MiB": np.full(10, 100 * MiB),
    "5000 x 16 KiB + 50 x 32 MiB": np.concatenate([np.full(5000, 16 * KiB), np.full(50, 32 * MiB)]),
    "30000 x 1 KiB": np.full(30000, KiB),
}


def best_of(c, build, tar, reps=5):
    c.tree(build, tar); c.tree(build, tar)
    rows = []
    for _ in range(reps):
        t0 = time.perf_counter(); y = c.tree(build, tar); dt = (time.perf_counter() - t0) * 1e3
        rows.append((dt, c.stats_ex()))
    rows.sort(key=lambda r: r[0])
    return rows[0][0], rows[len(rows) // 2][0], rows[0][1], y


tmp = tempfile.mkdtemp(prefix="snaphash_choice_", dir="/dev/shm")
try:
    blob = np.random.default_rng(1).integers(0, 256, size=(100 * MiB + 8192), dtype=np.uint8)
    for shape, sizes in SHAPES.items():
        if len(sys.argv) > 1 and not any(a in shape for a in sys.argv[1:]):
            continue
        root = os.path.join(tmp, "t")
        build = os.path.join(root, "build")
        for i, sz in enumerate(sizes):
            d = os.path.join(build, "d%04d" % (i // 100))
            if i % 100 == 0:
                os.makedirs(d)
            blob[i % 4096:(i % 4096) + int(sz)].tofile(os.path.join(d, "f%06d.bin" % i))
        tar = os.path.join(root, "data.tar.gz"); blob[:1000].tofile(tar)
        total = int(np.sum(sizes))
        with Context(flags=_lib.FLAG_GPU_ONLY) as c:
            g_best, g_med, _, y_gpu = best_of(c, build, tar)
        with Context() as c:
            d_best, d_med, ex, y_def = best_of(c, build, tar)
        assert y_gpu == y_def
        flag = "  <-- default slower than GPU-only" if d_med > 1.10 * g_med and d_med - g_med > 0.3 else ""
        print("%-44s %7.1f MiB | GPU only %8.2f / %8.2f ms | default %8.2f / %8.2f ms (host %d streams %d threads; planned gpu %.2f host %.2f, took gpu %.2f host %.2f)%s" % (
            shape, total / MiB, g_best, g_med, d_best, d_med, ex["host_streams"], ex["host_threads_run"], ex["planned_gpu_ms"], ex["planned_host_ms"],
            ex["gpu_ms"], ex["host_ms"], flag), flush=True)
        shutil.rmtree(root)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
