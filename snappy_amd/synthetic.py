"""Synthetic file trees of BASELINE.json's configs.

File i's content is the little-endian SplitMix64 stream seeded
0x5eed000000000000 ^ i; names d{i//100:04d}/f{i:06d}.bin.  These helpers only
describe trees (sizes, packed HBM offsets); the bytes are produced either on
the GPU (Context.fill_synthetic_device) or, for on-disk trees in tests, by the
pure-Python generator below.
"""
import struct

import numpy as np

M64 = (1 << 64) - 1
PACK_ALIGN = 256


def config_sizes(name):
    """-> numpy uint64 array of file sizes; the last entry is the data.tar stand-in."""
    if name == "C1":      # 100 x 64 KiB (+ archive)
        return np.full(101, 65536, dtype=np.uint64)
    if name == "C2":      # 10 000 x 1 MiB (+ 1 MiB archive stand-in)
        return np.full(10001, 1 << 20, dtype=np.uint64)
    if name == "C3":      # 100 x 1 GiB
        return np.full(100, 1 << 30, dtype=np.uint64)
    if name == "C5":      # Zipf 1 KiB .. 256 MiB, 100 000 files
        return zipf_sizes(100000)
    raise ValueError(name)


def zipf_sizes(n, seed=0xC5):
    """size(r) = clamp(floor(2^28 / r) - (r mod 113), 1 KiB, 256 MiB), rank r assigned
    to file indices by a seeded Fisher-Yates shuffle."""
    r = np.arange(1, n + 1, dtype=np.int64)
    size = np.clip((1 << 28) // r - (r % 113), 1024, 1 << 28).astype(np.uint64)
    rng = np.random.Generator(np.random.PCG64(seed))
    perm = rng.permutation(n)
    out = np.empty(n, dtype=np.uint64)
    out[perm] = size
    return out


def pack_offsets(lens, align=PACK_ALIGN):
    """Contiguous HBM layout: each file starts on an `align`-byte boundary."""
    lens = np.asarray(lens, dtype=np.uint64)
    padded = (lens + np.uint64(align - 1)) // np.uint64(align) * np.uint64(align)
    off = np.zeros(len(lens), dtype=np.uint64)
    if len(lens) > 1:
        off[1:] = np.cumsum(padded[:-1])
    total = int(off[-1] + padded[-1]) if len(lens) else 0
    return off, total


def file_bytes(length, index):
    """Pure-Python/numpy generator of one synthetic file (small sizes, tests)."""
    nwords = (length + 7) // 8
    j = np.arange(1, nwords + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(0x5EED000000000000 ^ index) + j * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.astype("<u8").tobytes()[:length]


def file_name(i):
    return "d%04d/f%06d.bin" % (i // 100, i)
