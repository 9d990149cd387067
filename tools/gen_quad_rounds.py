#!/usr/bin/env python3
"""Generator (and lane-level simulator) of the QUAD round block: the 80 SHA-512 rounds of one
128-byte block with every stream carried by FOUR lanes of one wave -- role (e-chain A / a-chain B,
as in the PAIR block) x half (low / high 32 bits of every 64-bit word).

Why: in the stream-starved regime a wave issues one instruction per ~4.4 cycles whatever it does, so
a stream's speed is the instruction count of the wave that holds its chaining value (DESIGN.md
sec. 4).  With the halves on separate lanes every bitwise operation and rotation is ONE instruction
instead of two; additions keep their carries in the upper register of a (value, 0) pair
(v_lshl_add_u64 on zero-extended halves) and hand them to the high lane once per round:
16 VALU per round instead of 19, no filler instructions (the DPP read-after-write distances are
covered by useful work), one ds_read_b32 per round.

Lane layout: a DPP row (16 lanes) = banks [A-lo | A-hi | B-lo | B-hi] of four streams; the high bank
is mirrored, so the other half of a lane's word is DPP row_half_mirror and the other role is DPP
row_ror:8.  bank_mask selects roles/halves: A 0x3, B 0xC, A-lo 0x1, B-lo 0x4, high lanes 0xA.

Per round (x0..x3 = the lane's four state words, rotating):
  TT = x3 + KW                         (A: h + K+W ; B: d + 0 -- B lanes read a zero word)
  P  = dpp_hm(x0) + SCR                the other half of x0: the high partner's value BEFORE its carry
                                       fix plus the carry this (low) lane holds; SCR is 0 in high lanes
  x0 += dpp_hm(SCR)   [high lanes]     the carry fix of the PREVIOUS round (F)
  T  = rot(x0, c1) ^ rot(x0, c2) ^ x0  two v_alignbit(P, x0, c) + one v_bitop3
  M  = x0 ^ (x2 & role); BF = bfi(M, x1, x2)      A: Ch(e,f,g)  B: Maj(a,b,c)
  S  = rot(T, c3)                      v_mov_dpp (other half of T) + v_alignbit
  VV = S + BF ; V2 = VV + TT           A: T1 ; B: T2 + d   (upper registers collect the carries)
  x3 = dpp_ror8(x3) + V2   [A]         e' = d + T1, carry -> SCR (low lanes only)
  x3 = dpp_ror8(V2) + VV   [B]         a' = T1 + T2, carries of both -> SCR (low lanes only)

The same instruction list is (a) printed as gfx950 assembly into snappy_amd/csrc/quad_rounds.inc and
(b) executed by `simulate()` on 64 numpy lanes (tests/test_quad_sim.py) -- the data flow is proven on
the CPU before it runs on the GPU.

  python3 tools/gen_quad_rounds.py          # rewrites quad_rounds.inc
"""
import os

import numpy as np

M32 = 0xFFFFFFFF
BASE = 64
WN = 4  # rounds of K+W per s_waitcnt (2 * WN register pairs)

REG = {}
def _alloc():
    n = BASE
    for name in ("R0", "R1", "R2", "R3", "HP0", "HP1", "HP2", "HP3", "S", "BF", "VV", "TT", "V2", "FF"):
        REG[name + "l"], REG[name + "h"] = n, n + 1
        n += 2
    for k in range(2 * WN):
        REG["KW%dl" % k], REG["KW%dh" % k] = n, n + 1
        n += 2
    for name in ("C1", "C2", "C3", "MB", "ADDR", "ZERO", "P", "T", "U", "PT", "M", "SCR"):
        REG[name] = n
        n += 1
    return n
LAST = _alloc() - 1

A_BANKS, B_BANKS, ALO, BLO, HI = 0x3, 0xC, 0x1, 0x4, 0xA
# registers that hold zero for the whole kernel: fixed-register inputs of the asm statement
ZERO_REGS = ["S" + "h", "BFh", "ZERO"] + ["KW%dh" % k for k in range(2 * WN)]
# in/out: the chaining words as (value, 0) pairs; constants; per-block LDS address
HP = ["HP0", "HP1", "HP2", "HP3"]
CONST_IN = ["C1", "C2", "C3", "MB", "ADDR"]


def build(rounds=80):
    ins = []
    e = ins.append
    e(("waitcnt", 0))
    for k in range(WN):
        e(("ds_read_b32", "KW%d" % k, "ADDR", k))
    for i in range(rounds):
        x = ["R%d" % ((k - i) % 4) for k in range(4)]
        kw = "KW%d" % (i % (2 * WN))
        if i % WN == 0:
            more = i + WN < rounds
            nxt = ((i // WN + 1) & 1) * WN
            if more:
                for k in range(WN):
                    e(("ds_read_b32", "KW%d" % (nxt + k), "ADDR", i + WN + k))
            e(("waitcnt", WN if more else 0))
            e(("nop",))
        e(("add64", "TT", x[3], kw))
        e(("add_dpp", "P", "hm", x[0] + "l", "SCR", 0xF))            # other half of x0 (see module docstring)
        if i > 0:
            e(("add_dpp", x[0] + "l", "hm", "SCR", x[0] + "l", HI))  # F of round i-1
        e(("alignbit", "T", "P", x[0] + "l", "C1"))
        e(("alignbit", "U", "P", x[0] + "l", "C2"))
        e(("bitop3", "T", "T", "U", x[0] + "l", 0x96))
        e(("bitop3", "M", x[0] + "l", x[2] + "l", "MB", 0x78))
        e(("bfi", "BFl", "M", x[1] + "l", x[2] + "l"))
        e(("mov_dpp", "PT", "hm", "T", 0xF))
        e(("alignbit", "Sl", "PT", "T", "C3"))
        e(("add64", "VV", "S", "BF"))
        e(("add64", "V2", "VV", "TT"))
        e(("add_co_dpp", x[3] + "l", "ror8", x[3] + "l", "V2l", A_BANKS))
        e(("addc_dpp", "SCR", "id", "ZERO", "V2h", ALO))
        e(("add_co_dpp", x[3] + "l", "ror8", "V2l", "VVl", B_BANKS))
        e(("addc_dpp", "SCR", "ror8", "V2h", "VVh", BLO))
    # F of the last round: the new x0 is R[(0 - rounds) % 4]
    x0 = "R%d" % ((0 - rounds) % 4)
    e(("nop8",))
    e(("nop8",))
    e(("add_dpp", x0 + "l", "hm", "SCR", x0 + "l", HI))
    return ins


def feed_forward():
    """After 80 rounds (register roles back at R0..R3): HPk <- HPk + Rk, in the (value, 0) pair format."""
    ins = []
    e = ins.append
    tmp = ["FF", "VV", "V2", "TT"]
    for k in range(4):
        e(("add64", tmp[k], "R%d" % k, "HP%d" % k))
    for k in range(4):
        e(("mov", "HP%dl" % k, tmp[k] + "l"))
    for k in range(4):
        e(("add_dpp", "HP%dl" % k, "hm", tmp[k] + "h", "HP%dl" % k, HI))
    return ins


# ---- assembly printer ----------------------------------------------------------------
def v(name):
    return "v%d" % REG[name]

def vp(name):
    return "v[%d:%d]" % (REG[name + "l"], REG[name + "h"])

DPP_CTRL = {"hm": "row_half_mirror", "ror8": "row_ror:8", "id": "quad_perm:[0,1,2,3]"}

def to_asm(ins):
    out = []
    for t in ins:
        op = t[0]
        if op == "waitcnt":
            out.append("s_waitcnt lgkmcnt(%d)" % t[1])
        elif op == "nop":
            out.append("s_nop 0")
        elif op == "nop8":  # an 8-byte no-op (keeps the stream at phase 0 mod 8): only in the block epilogue
            out.append("v_mov_b32_e64 %s, %s" % (v("PT"), v("PT")))
        elif op == "ds_read_b32":
            out.append("ds_read_b32 %s, %s offset:%d" % (v(t[1] + "l"), v(t[2]), 8 * t[3]))
        elif op == "alignbit":
            out.append("v_alignbit_b32 %s, %s, %s, %s" % (v(t[1]), v(t[2]), v(t[3]), v(t[4])))
        elif op == "bfi":
            out.append("v_bfi_b32 %s, %s, %s, %s" % (v(t[1]), v(t[2]), v(t[3]), v(t[4])))
        elif op == "bitop3":
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0x%x" % (v(t[1]), v(t[2]), v(t[3]), v(t[4]), t[5]))
        elif op == "add64":
            out.append("v_lshl_add_u64 %s, %s, 0, %s" % (vp(t[1]), vp(t[2]), vp(t[3])))
        elif op == "mov":
            out.append("v_mov_b32_e64 %s, %s" % (v(t[1]), v(t[2])))
        elif op == "mov_dpp":
            out.append("v_mov_b32_dpp %s, %s %s row_mask:0xf bank_mask:0x%x" % (v(t[1]), v(t[3]), DPP_CTRL[t[2]], t[4]))
        elif op == "add_dpp":
            out.append("v_add_u32_dpp %s, %s, %s %s row_mask:0xf bank_mask:0x%x" % (v(t[1]), v(t[3]), v(t[4]), DPP_CTRL[t[2]], t[5]))
        elif op == "add_co_dpp":
            out.append("v_add_co_u32_dpp %s, vcc, %s, %s %s row_mask:0xf bank_mask:0x%x" % (v(t[1]), v(t[3]), v(t[4]), DPP_CTRL[t[2]], t[5]))
        elif op == "addc_dpp":
            out.append("v_addc_co_u32_dpp %s, vcc, %s, %s, vcc %s row_mask:0xf bank_mask:0x%x" % (v(t[1]), v(t[3]), v(t[4]), DPP_CTRL[t[2]], t[5]))
        else:
            raise ValueError(op)
    return out


def write_inc(path):
    ins = build()
    body = to_asm(ins) + to_asm(feed_forward())
    n_valu = sum(1 for t in ins if t[0] not in ("waitcnt", "ds_read_b32", "nop"))
    lines = ["// GENERATED by tools/gen_quad_rounds.py -- do not edit.",
             "// One 128-byte block = 80 SHA-512 rounds with every stream on four lanes; %d VALU + %d LDS reads, then the" %
             (n_valu, sum(1 for t in ins if t[0] == "ds_read_b32")),
             "// feed-forward.  All registers are physical: the chaining words live in v[%d:%d] .. v[%d:%d] as (value, 0)" %
             (REG["HP0l"], REG["HP0h"], REG["HP3l"], REG["HP3h"]),
             "// pairs (in/out), constants and permanent zeros are fixed-register inputs (see SNAPHASH_QUAD_* below).",
             "#define SNAPHASH_QUAD_ROUNDS_ASM \\"]
    lines.append('    ".p2align 3\\n" \\')
    for k in range(4):  # working variables <- chaining words: v_mov_b64 copies (value, 0); 4 bytes each
        lines.append('    "v_mov_b64 %s, %s\\n" \\' % (vp("R%d" % k), vp("HP%d" % k)))
    lines.append('    "v_mov_b32_e32 %s, %s\\n" \\' % (v("SCR"), v("ZERO")))  # with the s_waitcnt that follows: 24 bytes
    for s_ in body:
        lines.append('    "%s\\n" \\' % s_)
    lines.append('    ""')
    def regs(names):
        return ", ".join('"v%d"' % REG[n] for n in names)
    internal = [n for n in REG if n not in ZERO_REGS and n not in CONST_IN and not n.startswith("HP")]
    lines.append("#define SNAPHASH_QUAD_CLOBBERS %s, \"vcc\"" % regs(sorted(internal, key=lambda n: REG[n])))
    for k in range(4):
        lines.append('#define SNAPHASH_QUAD_HP%d "+{v[%d:%d]}"' % (k, REG["HP%dl" % k], REG["HP%dh" % k]))
    for n in CONST_IN:
        lines.append('#define SNAPHASH_QUAD_%s "{v%d}"' % (n, REG[n]))
    lines.append("#define SNAPHASH_QUAD_ZERO_INPUTS(z) %s" % ", ".join('"{v%d}"(z)' % REG[n] for n in ZERO_REGS))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(body)


# ---- lane-level simulator ----------------------------------------------------------------
def lane_roles():
    """-> (is_b[64], is_hi[64], stream[64]) of the round wave's lane layout."""
    lane = np.arange(64)
    j = lane & 7
    is_hi = (lane & 4) != 0
    is_b = (lane & 8) != 0
    stream = 4 * (lane >> 4) + np.where(is_hi, 7 - j, j)
    return is_b, is_hi, stream

def dpp_source(ctrl):
    lane = np.arange(64)
    if ctrl == "hm":
        return (lane & ~7) | (7 - (lane & 7))
    if ctrl == "ror8":
        return (lane & ~15) | ((lane + 8) & 15)
    if ctrl == "id":
        return lane
    raise ValueError(ctrl)

def bank_enable(mask):
    lane = np.arange(64)
    return ((mask >> ((lane & 15) >> 2)) & 1).astype(bool)

def simulate(ins, regs, lds):
    """regs: dict name -> np.uint32[64]; lds: np.uint8 array.  Executes the list in program order; asserts
    that no register is read while an LDS read into it is outstanding and that DPP reads keep the two-
    instruction distance from the VALU write of their source (gfx9 hazard)."""
    pending = []
    vcc = np.zeros(64, dtype=np.uint64)
    written_at = {}   # register -> index of the instruction that wrote it (VALU writes only)
    def g(n):
        for p in pending:
            assert n != p, "read of %s before its ds_read was waited for" % n
        return regs[n]
    def g64(pair):
        return g(pair + "l").astype(np.uint64) | (g(pair + "h").astype(np.uint64) << np.uint64(32))
    def dpp_read(n, ctrl, idx):
        assert idx - written_at.get(n, -10) > 2, "DPP read of %s %d instruction(s) after its write" % (n, idx - written_at[n])
        return g(n)[dpp_source(ctrl)]
    def put(n, val, idx, en=None):
        regs[n] = val if en is None else np.where(en, val, regs[n])
        written_at[n] = idx
    for idx, t in enumerate(ins):
        op = t[0]
        if op in ("nop", "nop8"):
            continue
        if op == "waitcnt":
            while len(pending) > t[1]:
                pending.pop(0)
        elif op == "ds_read_b32":
            addr = g(t[2]).astype(np.int64) + 8 * t[3]
            val = np.zeros(64, dtype=np.uint32)
            for l in range(64):
                val[l] = int.from_bytes(bytes(lds[addr[l]:addr[l] + 4]), "little")
            regs[t[1] + "l"] = val
            pending.append(t[1] + "l")
        elif op == "alignbit":
            hi, lo, sh = g(t[2]).astype(np.uint64), g(t[3]).astype(np.uint64), (g(t[4]) & 31).astype(np.uint64)
            put(t[1], ((((hi << np.uint64(32)) | lo) >> sh) & np.uint64(M32)).astype(np.uint32), idx)
        elif op == "bfi":
            m = g(t[2])
            put(t[1], (m & g(t[3])) | (~m & g(t[4])), idx)
        elif op == "bitop3":
            a_, b_, c_ = g(t[2]), g(t[3]), g(t[4])
            r = np.zeros(64, dtype=np.uint32)
            for m in range(8):
                if t[5] & (1 << m):
                    r |= (a_ if m & 4 else ~a_) & (b_ if m & 2 else ~b_) & (c_ if m & 1 else ~c_)
            put(t[1], r, idx)
        elif op == "add64":
            with np.errstate(over="ignore"):
                r = g64(t[2]) + g64(t[3])
            put(t[1] + "l", (r & np.uint64(M32)).astype(np.uint32), idx)
            put(t[1] + "h", (r >> np.uint64(32)).astype(np.uint32), idx)
        elif op == "mov":
            put(t[1], g(t[2]).copy(), idx)
        elif op == "mov_dpp":
            put(t[1], dpp_read(t[3], t[2], idx), idx, bank_enable(t[4]))
        elif op == "add_dpp":
            s = dpp_read(t[3], t[2], idx).astype(np.uint64) + g(t[4]).astype(np.uint64)
            put(t[1], (s & np.uint64(M32)).astype(np.uint32), idx, bank_enable(t[5]))
        elif op in ("add_co_dpp", "addc_dpp"):
            en = bank_enable(t[5])
            s = dpp_read(t[3], t[2], idx).astype(np.uint64) + g(t[4]).astype(np.uint64)
            if op == "addc_dpp":
                s = s + vcc
            put(t[1], (s & np.uint64(M32)).astype(np.uint32), idx, en)
            vcc = np.where(en, s >> np.uint64(32), np.uint64(0xDEAD) & np.uint64(1))  # disabled lanes: not relied upon
        else:
            raise ValueError(op)
    assert not pending, "block ends with LDS reads in flight"
    return regs


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "snappy_amd", "csrc", "quad_rounds.inc")
    n = write_inc(out)
    print("wrote %s: %d instructions per block (%.2f per round), registers v%d..v%d" % (out, n, n / 80.0, BASE, LAST))
