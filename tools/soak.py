#!/usr/bin/env python3
"""Soak: the same C2 batch hashed N times per kernel variant; every digest vector must equal the
first (and the first is spot-checked against hashlib).  Catches rare hazard/ordering bugs that a
single passing run cannot.  usage: tools/soak.py [iterations]"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snappy_amd import Context, _lib, synthetic  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lens = synthetic.config_sizes("C2")
off, total = synthetic.pack_offsets(lens)
dev = torch.empty(total, dtype=torch.uint8, device="cuda")
ref = None
for name, k in (("pair", _lib.KERNEL_PAIR), ("split", _lib.KERNEL_SPLIT), ("wide", _lib.KERNEL_WIDE)):
    with Context(kernel=k) as c:
        if ref is None:
            c.fill_synthetic_device(dev.data_ptr(), off, lens, np.arange(len(lens), dtype=np.uint64))
        out = torch.zeros((len(lens), 64), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the ctx stream is non-blocking: order it after torch's memset
        n = iters if name == "pair" else max(20, iters // 6)
        bad = 0
        for it in range(n):
            out.zero_()
            c.sha512_device(dev.data_ptr(), off, lens, out.data_ptr())
            c.sync()
            if ref is None:
                ref = out.clone()
                for i in (0, 5000, 10000):
                    assert ref[i].cpu().numpy().tobytes() == hashlib.sha512(synthetic.file_bytes(int(lens[i]), i)).digest()
            elif not torch.equal(out, ref):
                bad += 1
        print("%-5s %4d iterations, %d mismatching vectors" % (name, n, bad), flush=True)
        assert bad == 0
print("soak ok")
