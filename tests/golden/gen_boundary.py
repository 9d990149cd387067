#!/usr/bin/env python3
"""Generate boundary_digests.json: hashlib.sha512 (OpenSSL) over the synthetic
SplitMix64 content at FIPS 180-4 boundary lengths.  Pure Python,
no project code imported, so the fixture is independent of oracle/ and of the
HIP path.  Run: python3 tests/golden/gen_boundary.py"""
import hashlib
import json
import os
import struct

M = (1 << 64) - 1
LENGTHS = [0, 1, 3, 55, 56, 63, 64, 111, 112, 113, 119, 120, 127, 128, 129, 239, 240, 241, 255, 256,
           257, 383, 384, 1000, 1023, 1024, 4096, 65535, 65536, 65537, 1048575, 1048576, 1048577]


def synthetic(length, index):
    s = 0x5EED000000000000 ^ index
    out = bytearray()
    while len(out) < length:
        s = (s + 0x9E3779B97F4A7C15) & M
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out += struct.pack("<Q", z)
    return bytes(out[:length])


def main():
    rows = []
    for i, n in enumerate(LENGTHS):
        data = synthetic(n, i)
        rows.append({"file_index": i, "length": n, "sha512": hashlib.sha512(data).hexdigest(),
                     "first8": data[:8].hex()})
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "boundary_digests.json"), "w") as f:
        json.dump({"generator": "splitmix64 seed 0x5eed000000000000^file_index, LE u64 stream, truncated",
                   "made_by": "tests/golden/gen_boundary.py (hashlib.sha512)", "vectors": rows}, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
