#!/usr/bin/env python3
"""The fused Build pass (bench.py end_to_end.build: tar + GPU DEFLATE + archive SHA-512 + per-file SHA-512 + hashes.yaml over 1 GiB of
Zipf-word text on tmpfs) at several efforts of the DEFLATE search: what the reference's gzip level (clickdeb/deb.go:271, level 9 =
depth 96's bytes) costs the pass.  usage: tools/build_depths.py [depths=32,64,96]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from snappy_amd import Context  # noqa: E402

bench.build_cpu_baselines = lambda *a, **k: {"skipped": "tools/build_depths.py"}
depths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,64,96").split(",")]
print(bench.bind_to_gpu_node(0))
for d in depths:
    with Context(deflate_depth=d) as c:
        r = bench.e2e_build(c, 0.0)
    print("depth %3d: build %.1f ms per GiB of text (%.2f GiB/s of tree), ratio %.4f, DEFLATE kernels %.1f ms, SHA-512 kernels %.1f ms" %
          (d, r["ms"], r["GiBps_of_tree"], r["ratio"], r["deflate_kernel_ms"], r["sha512_kernel_ms"]), flush=True)
