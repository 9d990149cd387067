#!/usr/bin/env python3
"""FIPS 180-4 sec. 4.2.3 / 5.3.5 constants from first principles: K[t] = first 64 bits of the
fractional part of the cube root of the t-th prime, IV[i] = likewise for the square root of the
i-th prime.  Prints the table used in snappy_amd/csrc/sha512_core.h (SNAPHASH_K512_LIST);
tests/test_core_host.py checks the header against `constants()`."""
from math import isqrt


def _icbrt(n):
    x = int(round(n ** (1.0 / 3)))
    while x ** 3 > n:
        x -= 1
    while (x + 1) ** 3 <= n:
        x += 1
    return x


def _primes(k):
    ps, n = [], 2
    while len(ps) < k:
        if all(n % p for p in ps):
            ps.append(n)
        n += 1
    return ps


def constants():
    ps = _primes(80)
    mask = (1 << 64) - 1
    return [_icbrt(p << 192) & mask for p in ps], [isqrt(p << 128) & mask for p in ps[:8]]


if __name__ == "__main__":
    K, IV = constants()
    for i in range(0, 80, 4):
        print("    " + ", ".join("0x%016xULL" % k for k in K[i:i + 4]) + ",")
    print("IV: " + ", ".join("0x%016xULL" % v for v in IV))
