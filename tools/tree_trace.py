#!/usr/bin/env python3
"""The N = 1 headline pass under the library's own trace: a C2-shaped tree (n x 1 MiB + archive) on tmpfs ->
snaphash_tree (GPU only), SNAPHASH_TRACE_TREE=1 prints walk / hash / yaml and the engine's wait / plan / read / enqueue /
drain sums per call.  usage: SNAPHASH_TRACE_TREE=1 tools/tree_trace.py [n=10000] [shards=1]
shards > 1: also what ONE rank of that many does (snaphash_shard_plan/_hash/_emit, rank 0), timed."""
import os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench  # noqa: E402  (bind_to_gpu_node)
from snappy_amd import Context, _lib, synthetic  # noqa: E402
from snappy_amd.sharded import ShardedTree  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print(bench.bind_to_gpu_node(0))
tmp = tempfile.mkdtemp(prefix="snaphash_tt_", dir="/dev/shm")
try:
    build = os.path.join(tmp, "build")
    blob = np.random.default_rng(3).integers(0, 256, size=(2 << 20), dtype=np.uint8)
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        blob[i % 4096:(i % 4096) + (1 << 20)].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    blob[:1 << 20].tofile(tar)
    total = (n + 1) << 20
    with Context(flags=_lib.FLAG_GPU_ONLY) as c:
        for rep in range(4):
            t0 = time.perf_counter()
            y = c.tree(build, tar)
            dt = time.perf_counter() - t0
            print("snaphash_tree: %.2f ms = %.2f GiB/s" % (dt * 1e3, total / 2**30 / dt), flush=True)
        if shards > 1:
            for rep in range(4):
                t0 = time.perf_counter()
                with ShardedTree(build, tar, 0, shards) as st:
                    t1 = time.perf_counter()
                    slab = st.hash(c)
                    t2 = time.perf_counter()
                    nbytes = st.bytes
                print("rank 0 of %d: plan %.2f ms, hash %.2f ms (%d streams, %.1f MiB = %.2f GiB/s of its own); a step of %d such ranks: >= %.2f ms = %.1f GiB/s of the tree" %
                      (shards, (t1 - t0) * 1e3, (t2 - t1) * 1e3, st.count, nbytes / 2**20, nbytes / 2**30 / (t2 - t1), shards, (t2 - t0) * 1e3,
                       total / 2**30 / (t2 - t0)), flush=True)
        if shards > 1:  # the shared walk (ABI 5), one rank's part of it timed here: its listing, then the plan from all ranks' listings
            import ctypes
            L = _lib.lib()
            blobs = []
            for r in range(shards):
                p, nb = ctypes.c_void_p(), ctypes.c_size_t()
                assert L.snaphash_shard_list(build.encode(), r, shards, ctypes.byref(p), ctypes.byref(nb)) == 0
                blobs.append(ctypes.string_at(p, nb.value))
                L.snaphash_free(p)
            keep = [ctypes.create_string_buffer(b, len(b)) for b in blobs]
            ptrs = (ctypes.c_void_p * shards)(*[ctypes.addressof(k) for k in keep])
            sizes = (ctypes.c_size_t * shards)(*[len(b) for b in blobs])
            for rep in range(4):
                t0 = time.perf_counter()
                p, nb = ctypes.c_void_p(), ctypes.c_size_t()
                assert L.snaphash_shard_list(build.encode(), 0, shards, ctypes.byref(p), ctypes.byref(nb)) == 0
                L.snaphash_free(p)
                t1 = time.perf_counter()
                h = ctypes.c_void_p()
                assert L.snaphash_shard_plan_from(build.encode(), tar.encode(), 0, shards, ptrs, sizes, ctypes.byref(h)) == 0
                t2 = time.perf_counter()
                L.snaphash_shard_free(h)
                print("rank 0 of %d, shared walk: its listing %.2f ms (%d bytes), the plan from all listings %.2f ms (+ two all-gathers of %d bytes in all)" %
                      (shards, (t1 - t0) * 1e3, len(blobs[0]), (t2 - t1) * 1e3, sum(len(b) for b in blobs)), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
