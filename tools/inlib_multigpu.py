#!/usr/bin/env python3
"""ONE process, every visible GPU, through the C ABI: what a Go `snappy build` reaches over cgo.
snaphash_init with the device list {-1} -> the library LPT-shards the file list over its engines (one
host thread + staging engine per device) and gathers the digest vector with a single-process RCCL
all-gather (FLAG_CHECK_GATHER: compared against per-device copies).  Host buffers in, digests out
(PCIe-inclusive).  Prints one JSON line.  Run by bench.py on rank 0 at N > 1 (in a child process with
a time limit, so that the headline line never depends on it) and usable on its own.
usage: tools/inlib_multigpu.py [C1|C2|C5] [devices, e.g. -1 or 0,0]"""
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device memory only: the synthetic tree is generated on GPU 0 and copied to the host)
from snappy_amd import Context, _lib, synthetic  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
devices = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "-1").split(",")]
sizes = synthetic.config_sizes(workload)
off, total = synthetic.pack_offsets(sizes)
torch.cuda.set_device(0)
with Context(device=0) as c0:
    dev = torch.empty(max(total, 16), dtype=torch.uint8, device="cuda")
    c0.fill_synthetic_device(dev.data_ptr(), off, sizes, np.arange(len(sizes), dtype=np.uint64))
    host = dev.cpu().numpy()
    del dev
    torch.cuda.empty_cache()
n = len(sizes)
ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + int(o) for o in off])
lens = (ctypes.c_uint64 * n)(*[int(x) for x in sizes])
out = ctypes.create_string_buffer(64 * n)
with Context(devices=devices, flags=_lib.FLAG_CHECK_GATHER | _lib.FLAG_GPU_ONLY) as c:
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        rc = _lib.lib().snaphash_sha512_buffers(c._h, ptrs, lens, n, out)
        dt = time.perf_counter() - t0
        if rc:
            raise SystemExit("snaphash_sha512_buffers failed: %d %s" % (rc, _lib.lib().snaphash_last_error(c._h)))
        if best is None or dt < best:
            best = dt
    ex, st = c.stats_ex(), c.stats()
    per_dev = [c.device_stats(i) for i in range(ex["n_devices"])]
digests = np.frombuffer(out.raw, dtype=np.uint8).reshape(n, 64)
rng = np.random.default_rng(3)
small = [int(i) for i in rng.integers(0, n, size=12) if sizes[int(i)] <= (64 << 20)] + [n - 1]
for i in small:
    assert digests[i].tobytes() == hashlib.sha512(host[int(off[i]):int(off[i]) + int(sizes[i])].tobytes()).digest(), i
print(json.dumps({
    "what": "single process, device list %s through the C ABI: host buffers -> in-library LPT shards -> per-device staging "
            "engines -> %s -> digests (PCIe-inclusive)" % (devices, {0: "no gather (one device)", 1: "single-process RCCL all-gather",
                                                                       2: "per-device copy gather"}[ex["gather_kind"]]),
    "workload": workload, "files": n, "bytes": int(sizes.sum()), "n_devices": ex["n_devices"], "gather_kind": ex["gather_kind"],
    "gather_checked_against_copies": bool(ex["gather_checked"]), "gather_ms": round(ex["gather_ms"], 3),
    "ms": round(best * 1e3, 2), "GiBps": round(float(sizes.sum()) / 2**30 / best, 2),
    "per_device_streams": [d["streams"] for d in per_dev], "per_device_kernel_ms": [round(d["kernel_ms"], 2) for d in per_dev],
    "per_device_h2d_ms": [round(d["h2d_ms"], 2) for d in per_dev],
    "parity": "bit-exact vs hashlib.sha512 on %d files" % len(small),
    "sha512_of_digest_vector": hashlib.sha512(digests.tobytes()).hexdigest()[:32]}))
