"""The PAIR round block (tools/gen_pair_rounds.py): the generated instruction
list is executed on 64 simulated lanes and checked against a plain SHA-512 round
function, and the emitted assembly is checked to be exactly that list.  CPU only;
the same list runs as gfx950 assembly in the -m gpu parity tests."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_pair_rounds as gp  # noqa: E402

M64 = (1 << 64) - 1


def rotr(x, n):
    return ((x >> n) | (x << (64 - n))) & M64


def ref_rounds(state, kw):
    a, b, c, d, e, f, g, h = state
    for t in range(len(kw)):
        S1 = rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41)
        ch = (e & f) ^ (~e & g & M64)
        t1 = (h + S1 + ch + kw[t]) & M64
        S0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39)
        mj = (a & b) ^ (a & c) ^ (b & c)
        t2 = (S0 + mj) & M64
        h, g, f, e = g, f, e, (d + t1) & M64
        d, c, b, a = c, b, a, (t1 + t2) & M64
    return [a, b, c, d, e, f, g, h]


def run_block(rounds, seed, loop_rounds=0):
    rng = np.random.default_rng(seed)
    is_b, stream = gp.lane_roles()
    assert sorted(stream.tolist()) == sorted(list(range(32)) * 2)
    P = gp.partner_index()
    assert (stream[P] == stream).all() and (is_b[P] != is_b).all()  # the partner is the other role of the same stream
    states = [[int(x) for x in rng.integers(0, 1 << 63, size=8, dtype=np.uint64) * 2 + rng.integers(0, 2, size=8, dtype=np.uint64)]
              for _ in range(32)]
    kws = [[int(x) for x in rng.integers(0, 1 << 63, size=rounds, dtype=np.uint64) * 2 + 1] for _ in range(32)]
    row_bytes = 81 * 8
    lds = np.zeros(4096 + 32 * row_bytes, dtype=np.uint8)
    base = 1024
    for s in range(32):
        for t in range(rounds):
            lds[base + s * row_bytes + 8 * t: base + s * row_bytes + 8 * t + 8] = np.frombuffer(
                kws[s][t].to_bytes(8, "little"), dtype=np.uint8)
    regs = {n: rng.integers(0, 1 << 32, size=64, dtype=np.uint64).astype(np.uint32) for n in gp.REG}  # junk everywhere
    for lane in range(64):
        st = states[stream[lane]]
        mine = st[0:4] if is_b[lane] else st[4:8]     # B: a,b,c,d ; A: e,f,g,h
        for k in range(4):
            regs["R%dl" % k][lane] = mine[k] & 0xFFFFFFFF
            regs["R%dh" % k][lane] = mine[k] >> 32
        c = (6, 11, 28) if is_b[lane] else (4, 27, 14)
        regs["C1"][lane], regs["C2"][lane], regs["C3"][lane] = c
        regs["MB"][lane] = 0xFFFFFFFF if is_b[lane] else 0
        regs["ADDR"][lane] = base + stream[lane] * row_bytes
    out = gp.simulate(gp.build(rounds, loop_rounds), regs, lds)
    for lane in range(64):
        want = ref_rounds(states[stream[lane]], kws[stream[lane]])
        want = want[0:4] if is_b[lane] else want[4:8]
        rot = rounds % 4  # register roles rotate by one per round
        for k in range(4):
            r = "R%d" % ((k - rot) % 4)
            got = int(out[r + "l"][lane]) | (int(out[r + "h"][lane]) << 32)
            assert got == want[k], (rounds, lane, k)


def test_pair_rounds_match_sha512():
    for rounds, seed in ((2, 2), (4, 4), (6, 5), (80, 6), (80, 7)):
        run_block(rounds, seed)
    for loop_rounds, seed in ((4, 8), (8, 9), (16, 10), (gp.LOOP_ROUNDS, 11)):
        run_block(80, seed, loop_rounds)  # hardware-loop forms, incl. the shipped one


def test_sigma_decomposition():
    # Sigma1(e) = rotr14(e ^ rotr4 e ^ rotr27 e), Sigma0(a) = rotr28(a ^ rotr6 a ^ rotr11 a): what lets one
    # instruction stream serve both roles with per-lane rotate amounts and no half-swap
    rng = np.random.default_rng(0)
    for x in [int(v) for v in rng.integers(0, 1 << 63, size=50, dtype=np.uint64)]:
        assert rotr(x ^ rotr(x, 4) ^ rotr(x, 27), 14) == rotr(x, 14) ^ rotr(x, 18) ^ rotr(x, 41)
        assert rotr(x ^ rotr(x, 6) ^ rotr(x, 11), 28) == rotr(x, 28) ^ rotr(x, 34) ^ rotr(x, 39)


def test_generated_inc_is_current():
    """snappy_amd/csrc/pair_rounds.inc must be exactly what the generator prints."""
    import tempfile
    path = os.path.join(ROOT, "snappy_amd", "csrc", "pair_rounds.inc")
    with tempfile.TemporaryDirectory() as tmp:
        fresh = os.path.join(tmp, "x.inc")
        n = gp.write_inc(fresh)
        assert open(fresh).read() == open(path).read()
    if gp.LOOP_ROUNDS:
        L = gp.LOOP_ROUNDS  # body: L rounds of 19 VALU, L/2 (ds_read2_b64 + s_waitcnt); + loop control
        assert n == 2 + 2 + L * 19 + L + 1 + 3 + 1
    else:
        # per round 19 VALU; per eight rounds four ds_read2_b64, one s_waitcnt and one s_nop 0 (keeps the stream 8-byte aligned)
        assert n == 1 + 80 * 19 + 40 + 10 + 10


def test_dpp_hazard_distance():
    """gfx9: a VGPR written by VALU must not be read through DPP by either of the next two instructions."""
    ins = [t for t in gp.build(loop_rounds=0) if t[0] != "waitcnt"]
    for i, t in enumerate(ins):
        if t[0] in ("add_co_dpp", "addc_co_dpp"):
            src = t[2]
            for back in (1, 2):
                p = ins[i - back]
                written = {p[1]} if p[0] not in ("add64", "ds_read2_b64") else {p[1] + "l", p[1] + "h"}
                assert src not in written, (i, t, p)
