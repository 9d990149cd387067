"""The QUAD round block (tools/gen_quad_rounds.py: role x half, four lanes per stream): the generated
instruction list is executed on 64 simulated lanes -- DPP row_half_mirror / row_ror:8 with bank masks,
carries kept in (value, 0) pairs -- and checked against a plain SHA-512 round function; the simulator also
enforces the gfx9 DPP read-after-write distance and the lgkmcnt discipline.  CPU only; the same list runs as
gfx950 assembly in the -m gpu parity tests (kernel variant "quad")."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_quad_rounds as gq  # noqa: E402
from test_pair_sim import ref_rounds, M64  # noqa: E402


def run_block(rounds, seed, with_ff=False):
    rng = np.random.default_rng(seed)
    is_b, is_hi, stream = gq.lane_roles()
    assert sorted(stream.tolist()) == sorted(list(range(16)) * 4)
    hm, ror = gq.dpp_source("hm"), gq.dpp_source("ror8")
    assert (stream[hm] == stream).all() and (is_hi[hm] != is_hi).all() and (is_b[hm] == is_b).all()
    assert (stream[ror] == stream).all() and (is_b[ror] != is_b).all() and (is_hi[ror] == is_hi).all()
    states = [[int(x) for x in rng.integers(0, 1 << 63, size=8, dtype=np.uint64) * 2 + rng.integers(0, 2, size=8, dtype=np.uint64)]
              for _ in range(16)]
    kws = [[int(x) for x in rng.integers(0, 1 << 63, size=rounds, dtype=np.uint64) * 2 + 1] for _ in range(16)]
    row_bytes = 81 * 8
    base, zero_at = 1024, 256  # K+W rows; a zero region for the B lanes
    lds = np.zeros(base + 16 * row_bytes + 64, dtype=np.uint8)
    for s in range(16):
        for t in range(rounds):
            lds[base + s * row_bytes + 8 * t: base + s * row_bytes + 8 * t + 8] = np.frombuffer(
                kws[s][t].to_bytes(8, "little"), dtype=np.uint8)
    regs = {n: rng.integers(0, 1 << 32, size=64, dtype=np.uint64).astype(np.uint32) for n in gq.REG}  # junk everywhere
    for n in gq.ZERO_REGS + ["SCR"] + ["R%dh" % k for k in range(4)] + ["HP%dh" % k for k in range(4)]:
        regs[n] = np.zeros(64, dtype=np.uint32)
    for lane in range(64):
        st = states[stream[lane]]
        mine = st[0:4] if is_b[lane] else st[4:8]     # B: a,b,c,d ; A: e,f,g,h
        for k in range(4):
            half = (mine[k] >> 32) if is_hi[lane] else (mine[k] & 0xFFFFFFFF)
            regs["R%dl" % k][lane] = half
            regs["HP%dl" % k][lane] = half
        c = (6, 11, 28) if is_b[lane] else (4, 27, 14)
        regs["C1"][lane], regs["C2"][lane], regs["C3"][lane] = c
        regs["MB"][lane] = 0xFFFFFFFF if is_b[lane] else 0
        regs["ADDR"][lane] = zero_at if is_b[lane] else base + stream[lane] * row_bytes + (4 if is_hi[lane] else 0)
    ins = gq.build(rounds) + (gq.feed_forward() if with_ff else [])
    out = gq.simulate(ins, regs, lds)
    for lane in range(64):
        st = states[stream[lane]]
        want = ref_rounds(st, kws[stream[lane]])
        if with_ff:
            want = [(w + s) & M64 for w, s in zip(want, st)]
        want = want[0:4] if is_b[lane] else want[4:8]
        rot = rounds % 4
        for k in range(4):
            r = ("HP%d" % k) if with_ff else "R%d" % ((k - rot) % 4)
            got = int(out[r + "l"][lane])
            w = (want[k] >> 32) if is_hi[lane] else (want[k] & 0xFFFFFFFF)
            assert got == w, (rounds, lane, k, with_ff)
            assert int(out[r + "h"][lane]) == 0  # the upper register of a state pair stays zero
    for n in gq.ZERO_REGS:
        assert not out[n].any(), n
    assert not out["SCR"][is_hi].any()  # the carry register is never written in high lanes


def test_quad_rounds_match_sha512():
    for rounds, seed in ((1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (9, 6), (80, 7), (80, 8)):
        run_block(rounds, seed)
    run_block(80, 9, with_ff=True)
    run_block(80, 10, with_ff=True)


def test_generated_inc_is_current():
    import tempfile
    path = os.path.join(ROOT, "snappy_amd", "csrc", "quad_rounds.inc")
    with tempfile.TemporaryDirectory() as tmp:
        fresh = os.path.join(tmp, "x.inc")
        n = gq.write_inc(fresh)
        assert open(fresh).read() == open(path).read()
    # per round 16 VALU + 1 ds_read_b32 (the first round has no carry fix: it is the last instruction of the
    # block instead); per four rounds one s_waitcnt + one s_nop 0; two 8-byte no-ops in front of the last fix;
    # the feed-forward is 12 instructions
    assert n == 1 + 80 * 17 + 20 * 2 + 2 + 12


def test_every_instruction_is_eight_bytes_except_paired_waits():
    """Code placement (profiles/r02_pair_alignment.txt): the stream must stay at phase 0 mod 8, so the only
    4-byte instructions allowed in the round stream are s_waitcnt immediately followed by s_nop 0."""
    asm = gq.to_asm(gq.build())
    for i, line in enumerate(asm[5:], 5):
        if line.startswith("s_waitcnt"):
            assert asm[i + 1] == "s_nop 0", i
        elif line == "s_nop 0":
            assert asm[i - 1].startswith("s_waitcnt"), i
        else:
            assert line.split()[0] in ("v_lshl_add_u64", "v_alignbit_b32", "v_bitop3_b32", "v_bfi_b32", "ds_read_b32",
                                       "v_mov_b32_e64") or "_dpp" in line.split()[0], line
