import torch, time
n = 256 << 20
hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
ds = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
for streams in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(8):
            for k in range(streams):
                with torch.cuda.stream(ss[k]):
                    ds[k].copy_(hs[k], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d concurrent H2D streams of 256 MiB copies: %.1f GB/s" % (streams, 8 * streams * n / dt / 1e9))
for sz in (8 << 20, 32 << 20, 64 << 20):
    s = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for it in range(64):
            ds[0][:sz].copy_(hs[0][:sz], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("back-to-back copies of %d MiB on one stream: %.1f GB/s" % (sz >> 20, 64 * sz / dt / 1e9))
