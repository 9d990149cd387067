#!/usr/bin/env python3
"""The many-stream kernel in its forms (SNAPHASH_WIDE_FORM: 0 = LDS-staged tile, 3 waves per SIMD; 1/2/3 = direct
per-lane loads at 5/6/8 waves per SIMD), kernel-resident, equal streams; digests compared across forms.
usage: tools/wide_forms.py            (parent: runs itself once per form)
       tools/wide_forms.py child      (one form, from the environment)"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [(16384, 512 << 10), (32768, 256 << 10), (65536, 128 << 10), (131072, 128 << 10), (262144, 64 << 10), (100000, 40000 + 77)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    from snappy_amd import Context, _lib, synthetic
    with Context(kernel=_lib.KERNEL_WIDE) as c:
        for n, size in CASES:
            lens = np.full(n, size, dtype=np.uint64)
            off, total = synthetic.pack_offsets(lens)
            dev = torch.empty(total + 4096, dtype=torch.uint8, device="cuda")
            out = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
            c.fill_synthetic_device(dev.data_ptr(), off, lens, np.arange(n, dtype=np.uint64))
            ms = []
            for rep in range(4):
                c.sha512_device(dev.data_ptr(), off, lens, out.data_ptr())
                c.sync()
                ms.append(c.stats()["kernel_ms"])
            h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
            print("form %s  %7d x %7d B: %8.2f ms = %7.1f GB/s  digests %s" % (os.environ.get("SNAPHASH_WIDE_FORM", "0"), n, size, min(ms),
                                                                            n * size / min(ms) / 1e6, h), flush=True)
            del dev, out
            torch.cuda.empty_cache()
else:
    for form in (sys.argv[1:] or ["0", "1", "2", "3"]):
        env = dict(os.environ, SNAPHASH_WIDE_FORM=form)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
