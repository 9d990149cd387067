#!/usr/bin/env python3
"""PCIe H2D rate from pinned memory: one stream vs the same bytes split over two/four streams (does splitting the
staging engine's copies pay?).  usage: tools/h2d_probe.py"""
import time

import torch

N = 256 << 20
host = torch.empty(N, dtype=torch.uint8).pin_memory()
dev = torch.empty(N, dtype=torch.uint8, device="cuda")
for parts in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    best = 1e9
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(8):
            for k, s in enumerate(streams):
                a, b = k * N // parts, (k + 1) * N // parts
                with torch.cuda.stream(s):
                    dev[a:b].copy_(host[a:b], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 8)
    print("H2D %d MiB in %d concurrent part(s): %.2f ms = %.1f GB/s" % (N >> 20, parts, best * 1e3, N / best / 1e9), flush=True)
