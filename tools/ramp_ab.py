#!/usr/bin/env python3
"""A/B of the staging engine's lab knobs INSIDE one process: a C2-shaped tree on tmpfs -> snaphash_tree (GPU only), the
settings alternated call by call (the knobs are read once a call), so that box-to-box and minute-to-minute noise hits
every setting alike.  usage: tools/ramp_ab.py [n=10000] [rounds=8] "KNOB=v;KNOB=v" ...   (SNAPHASH_ is prefixed; "" = the defaults), e.g.
   "NEW_PER_BATCH=0;RAMP_MANY=8,100;HOLD_BACK=0" (rounds 1-4's engine)  "" (today's)
RAMP_MANY = first batch in 1/64ths of a full one, growth per batch in percent ("x,100" = doubling from 1/8).
Prints per setting: min / median of the pass, and of one rank's step of eight (plan + hash)."""
import os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench  # noqa: E402
from snappy_amd import Context, _lib, synthetic  # noqa: E402
from snappy_amd.sharded import ShardedTree  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
settings = sys.argv[3:] or ["NEW_PER_BATCH=0;RAMP_MANY=8,100;HOLD_BACK=0", ""]
KNOBS = ("NEW_PER_BATCH", "RAMP_MANY", "HOLD_BACK", "RAMP_SHIFT")


def apply(setting):
    for k in KNOBS:
        os.environ.pop("SNAPHASH_" + k, None)
    for kv in filter(None, setting.split(";")):
        k, v = kv.split("=")
        assert k in KNOBS, k
        os.environ["SNAPHASH_" + k] = v
print(bench.bind_to_gpu_node(0))
tmp = tempfile.mkdtemp(prefix="snaphash_ab_", dir="/dev/shm")
try:
    build = os.path.join(tmp, "build")
    blob = np.random.default_rng(3).integers(0, 256, size=(2 << 20), dtype=np.uint8)
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(i))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        blob[i % 4096:(i % 4096) + (1 << 20)].tofile(p)
    tar = os.path.join(tmp, "data.tar.gz")
    blob[:1 << 20].tofile(tar)
    tree = {s: [] for s in settings}
    rank = {s: [] for s in settings}
    with Context(flags=_lib.FLAG_GPU_ONLY) as c:
        ref = c.tree(build, tar)
        c.tree(build, tar)
        for r in range(rounds):
            for s in settings:
                apply(s)
                t0 = time.perf_counter()
                y = c.tree(build, tar)
                tree[s].append((time.perf_counter() - t0) * 1e3)
                assert y == ref
                t0 = time.perf_counter()
                with ShardedTree(build, tar, 0, 8) as st:
                    st.hash(c)
                rank[s].append((time.perf_counter() - t0) * 1e3)
    for s in settings:
        print("%-52s: tree min %.2f median %.2f ms | rank of 8 (plan + hash) min %.2f median %.2f ms" %
              (s or "(defaults)", min(tree[s]), sorted(tree[s])[len(tree[s]) // 2], min(rank[s]), sorted(rank[s])[len(rank[s]) // 2]))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
