#!/usr/bin/env python3
"""Generator (and lane-level simulator) of the PAIR round block: the 80 SHA-512
rounds of one 128-byte block with every stream carried by a lane PAIR.

Why: in the stream-starved regime a wave issues ~5 cycles per instruction
whatever it does, so a stream's speed is the instruction count of the wave
holding its chaining value.  Splitting a stream over two
lanes of the same wave -- lane A owns (e,f,g,h) and computes T1, lane B owns
(a,b,c,d) and computes T2 -- lets ONE instruction serve Sigma1(e) in the A lanes
and Sigma0(a) in the B lanes (per-lane rotate amounts in a VGPR), and likewise
Ch/Maj and the adds.  With CDNA4's v_bitop3_b32 (three-input truth-table op) the three-way
xor of the Sigma functions and M = x0 ^ (x2 & role) are one instruction per half:
19 instructions per round (33 - 6 on a single lane with the same op).

Lane layout (64 lanes = 32 streams): in every group of 8 lanes, lanes 0-3 are
role A of four streams and lanes 4-7 role B of the same streams mirrored, so
the partner of a lane is DPP row_half_mirror and DPP bank_mask 0x5 / 0xA writes
only A / only B lanes.

The same instruction list is (a) printed as gfx950 assembly into
snappy_amd/csrc/pair_rounds.inc (physical VGPRs, used inside one inline-asm
statement by sha512_kernels.hip) and (b) executed by `simulate()` on 64 numpy
lanes, which tests/test_pair_sim.py checks against a plain SHA-512 -- the data
flow is proven on the CPU before it ever runs on the GPU.

  python3 tools/gen_pair_rounds.py          # rewrites pair_rounds.inc
"""
import os

import numpy as np

M32 = 0xFFFFFFFF
BASE = 64  # first physical VGPR of the block; the kernel lists v64..v157 as clobbers

# ---- register map -----------------------------------------------------------------
REG = {}
def _alloc():
    n = BASE
    for name in ("R0", "R1", "R2", "R3"):          # the four rotating state pairs
        REG[name + "l"], REG[name + "h"] = n, n + 1
        n += 2
    for name in ("C1", "C2", "C3", "MB", "ADDR", "spare"):  # per-lane constants
        REG[name] = n
        n += 1
    for name in ["T", "U", "S", "M", "BF", "VV", "TT", "V2"] + ["KW%d" % k for k in range(32)]:  # aligned pairs
        REG[name + "l"], REG[name + "h"] = n, n + 1
        n += 2
    return n
LAST = _alloc() - 1
assert BASE % 2 == 0 and LAST == 157

IN_ORDER = ["R0l", "R0h", "R1l", "R1h", "R2l", "R2h", "R3l", "R3h", "C1", "C2", "C3", "MB", "ADDR"]
A_BANKS, B_BANKS = 0x5, 0xA


# Code placement experiment (MI355X_MICROARCH.md, "Code-placement sensitivity of hand-written streams"): every
# instruction of the block is 8 bytes except s_waitcnt (4), so the stream's phase mod 8 flips at every wait.
#   "none": as is;  "nop": an s_nop 0 behind every in-round s_waitcnt (phase constant, one more issue slot per
#   two rounds);  "p2": .p2align 3 in front of the block;  "both".
ALIGN = os.environ.get("SNAPHASH_PAIR_ALIGN", "w8")
# "w4" (shipped): K+W arrives four rounds per wait -- two ds_read2_b64 back to back every four rounds, ONE
# s_waitcnt + ONE s_nop 0 (together 8 bytes) in front of the first use: the instruction stream stays at phase 0
# mod 8 throughout at the same number of issue slots as the two-round scheme.  Measured on C2 (10 001 x 1 MiB):
# none 26.9 ms, nop 25.3 ms, w4 see profiles/r02_pair_alignment.txt.
SCHEME = "w4" if ALIGN in ("w4", "w8", "w16") else "w2"
WN = {"w8": 8, "w16": 16}.get(ALIGN, 4)  # rounds of K+W per wait in the aligned scheme (2 * WN register pairs)
_active_scheme = [SCHEME]  # the hardware-loop forms (experiments) always use the two-round scheme
ORDER = os.environ.get("SNAPHASH_PAIR_ORDER", "interleave")  # "interleave" (shipped: VOP2 between the VOP3s, 30.06 vs 30.20 ms on C2) or "plain"


def one_round(e, i, kw_prefetch, wait):
    """Appends round i (register roles depend on i mod 4 only).  kw_prefetch: None, a K+W word index (two-round
    scheme: one ds_read2_b64 every two rounds) or a list of (register, word index) reads (four-round scheme)."""
    x = ["R%d" % ((k - i) % 4) for k in range(4)]  # x0..x3 of this round
    kw = "KW%d" % ((i % (2 * WN)) if _active_scheme[0] == "w4" else (i & 3))
    if isinstance(kw_prefetch, list):
        for reg, word in kw_prefetch:
            e(("ds_read2_b64", reg, "ADDR", word))
    elif kw_prefetch is not None:
        e(("ds_read2_b64", "KW%d" % ((i + 2) & 3), "ADDR", kw_prefetch))
    if ORDER.startswith("seed:"):
        # experiment: a random topological order of the round's 15 non-DPP instructions (the four
        # DPP adds stay last, in order); dependencies are read off the operand names
        import random
        body = [("alignbit", "Tl", x[0] + "h", x[0] + "l", "C1"), ("alignbit", "Th", x[0] + "l", x[0] + "h", "C1"),
                ("alignbit", "Ul", x[0] + "h", x[0] + "l", "C2"), ("alignbit", "Uh", x[0] + "l", x[0] + "h", "C2"),
                ("bitop3", "Tl", "Tl", "Ul", x[0] + "l", 0x96), ("bitop3", "Th", "Th", "Uh", x[0] + "h", 0x96),
                ("alignbit", "Sl", "Th", "Tl", "C3"), ("alignbit", "Sh", "Tl", "Th", "C3"),
                ("bitop3", "Ml", x[0] + "l", x[2] + "l", "MB", 0x78), ("bitop3", "Mh", x[0] + "h", x[2] + "h", "MB", 0x78),
                ("bfi", "BFl", "Ml", x[1] + "l", x[2] + "l"),
                ("bfi", "BFh", "Mh", x[1] + "h", x[2] + "h"), ("add64", "VV", "S", "BF"), ("add64", "TT", x[3], kw),
                ("add64", "V2", "VV", "TT")]
        def regs_of(t, write):
            names = [t[1]] if write else [q for q in t[2:] if isinstance(q, str)]
            out = set()
            for nme in names:
                out |= {nme + "l", nme + "h"} if t[0] == "add64" else {nme}
            return out
        rnd = random.Random(int(ORDER[5:]) * 1000 + (i & 3))
        done, left = [], list(range(len(body)))
        while left:
            ready = []
            for k in left:  # all earlier (program-order) instructions that conflict with k must be done
                ok = True
                for j in range(k):
                    if j in left:
                        wj, rj = regs_of(body[j], True), regs_of(body[j], False)
                        wk, rk = regs_of(body[k], True), regs_of(body[k], False)
                        if (wj & rk) or (wj & wk) or (rj & wk):
                            ok = False
                            break
                if ok:
                    ready.append(k)
            k = rnd.choice(ready)
            left.remove(k)
            done.append(k)
        for k in done:
            if body[k][0] == "add64" and body[k][1] == "TT" and wait is not None:
                e(("waitcnt", wait))
                if ALIGN in ("nop", "both", "w4", "w8", "w16"):
                    e(("nop",))
            e(body[k])
        e(("add_co_dpp", x[3] + "l", x[3] + "l", "V2l", A_BANKS))
        e(("addc_co_dpp", x[3] + "h", x[3] + "h", "V2h", A_BANKS))
        e(("add_co_dpp", x[3] + "l", "V2l", "VVl", B_BANKS))
        e(("addc_co_dpp", x[3] + "h", "V2h", "VVh", B_BANKS))
        return
    if ORDER == "interleave":
        e(("alignbit", "Tl", x[0] + "h", x[0] + "l", "C1"))
        e(("alignbit", "Th", x[0] + "l", x[0] + "h", "C1"))
        e(("alignbit", "Ul", x[0] + "h", x[0] + "l", "C2"))
        e(("alignbit", "Uh", x[0] + "l", x[0] + "h", "C2"))
        e(("bitop3", "Ml", x[0] + "l", x[2] + "l", "MB", 0x78))
        e(("bitop3", "Mh", x[0] + "h", x[2] + "h", "MB", 0x78))
        e(("bitop3", "Tl", "Tl", "Ul", x[0] + "l", 0x96))
        e(("bitop3", "Th", "Th", "Uh", x[0] + "h", 0x96))
        e(("bfi", "BFl", "Ml", x[1] + "l", x[2] + "l"))
        e(("bfi", "BFh", "Mh", x[1] + "h", x[2] + "h"))
        if wait is not None:
            e(("waitcnt", wait))
            if ALIGN in ("nop", "both", "w4", "w8", "w16"):
                e(("nop",))
        e(("add64", "TT", x[3], kw))
        e(("alignbit", "Sl", "Th", "Tl", "C3"))
        e(("alignbit", "Sh", "Tl", "Th", "C3"))
        e(("add64", "VV", "S", "BF"))
        e(("add64", "V2", "VV", "TT"))
        e(("add_co_dpp", x[3] + "l", x[3] + "l", "V2l", A_BANKS))
        e(("addc_co_dpp", x[3] + "h", x[3] + "h", "V2h", A_BANKS))
        e(("add_co_dpp", x[3] + "l", "V2l", "VVl", B_BANKS))
        e(("addc_co_dpp", x[3] + "h", "V2h", "VVh", B_BANKS))
        return
    # S = rotr(x0 ^ rotr(x0,c1) ^ rotr(x0,c2), c3)   [A: Sigma1(e) c=(4,27,14); B: Sigma0(a) c=(6,11,28)]
    e(("alignbit", "Tl", x[0] + "h", x[0] + "l", "C1"))
    e(("alignbit", "Th", x[0] + "l", x[0] + "h", "C1"))
    e(("alignbit", "Ul", x[0] + "h", x[0] + "l", "C2"))
    e(("alignbit", "Uh", x[0] + "l", x[0] + "h", "C2"))
    e(("bitop3", "Tl", "Tl", "Ul", x[0] + "l", 0x96))   # three-way xor
    e(("bitop3", "Th", "Th", "Uh", x[0] + "h", 0x96))
    e(("alignbit", "Sl", "Th", "Tl", "C3"))
    e(("alignbit", "Sh", "Tl", "Th", "C3"))
    # M = x0 ^ (x2 & MB)   [A: e ; B: a ^ c];  BF = bfi(M, x1, x2)   [A: Ch(e,f,g) ; B: Maj(a,b,c)]
    e(("bitop3", "Ml", x[0] + "l", x[2] + "l", "MB", 0x78))
    e(("bitop3", "Mh", x[0] + "h", x[2] + "h", "MB", 0x78))
    e(("bfi", "BFl", "Ml", x[1] + "l", x[2] + "l"))
    e(("bfi", "BFh", "Mh", x[1] + "h", x[2] + "h"))
    e(("add64", "VV", "S", "BF"))            # A: Sigma1+Ch ; B: T2 = Sigma0+Maj
    if wait is not None:
        e(("waitcnt", wait))
        if ALIGN in ("nop", "both", "w4", "w8", "w16"):
            e(("nop",))
    e(("add64", "TT", x[3], kw))             # A: h + (K+W) ; B: unused
    e(("add64", "V2", "VV", "TT"))           # A: T1 ; B: unused
    # x3 <- new chain value (x0 of the next round)
    e(("add_co_dpp", x[3] + "l", x[3] + "l", "V2l", A_BANKS))    # A: e' = d(partner) + T1
    e(("addc_co_dpp", x[3] + "h", x[3] + "h", "V2h", A_BANKS))
    e(("add_co_dpp", x[3] + "l", "V2l", "VVl", B_BANKS))         # B: a' = T1(partner) + T2
    e(("addc_co_dpp", x[3] + "h", "V2h", "VVh", B_BANKS))


def build(rounds=80, loop_rounds=0):
    """-> list of instruction tuples for one block.
    K+W arrives two rounds per LDS instruction, two rounds ahead of its first use: register
    sets (KW0,KW1) and (KW2,KW3) alternate every two rounds.
    loop_rounds = 0: fully unrolled.  loop_rounds = L (multiple of 4, divides rounds): a
    hardware loop of rounds/L iterations over an L-round body; ADDR advances 8*L bytes per
    iteration and the last iteration prefetches two words past the row (never used)."""
    ins = []
    e = ins.append
    _active_scheme[0] = SCHEME if not loop_rounds else "w2"
    if SCHEME == "w4" and not loop_rounds:
        e(("waitcnt", 0))
        for k in range(0, WN, 2):
            e(("ds_read2_b64", "KW%d" % k, "ADDR", k))
        for i in range(rounds):
            if i % WN == 0:
                more = i + WN < rounds
                nxt = ((i // WN + 1) & 1) * WN
                pre = [("KW%d" % (nxt + k), i + WN + k) for k in range(0, WN, 2)] if more else []
                one_round(e, i, pre, WN // 2 if more else 0)
            else:
                one_round(e, i, None, None)
        return ins
    e(("waitcnt", 0))
    e(("ds_read2_b64", "KW0", "ADDR", 0))
    if not loop_rounds:
        for i in range(rounds):
            more = (i & 1) == 0 and i + 2 < rounds
            one_round(e, i, (i + 2) if more else None, (1 if more else 0) if (i & 1) == 0 else None)
        return ins
    assert loop_rounds % 4 == 0 and rounds % loop_rounds == 0
    e(("loop_begin", rounds // loop_rounds))
    for i in range(loop_rounds):
        even = (i & 1) == 0
        one_round(e, i, (i + 2) if even else None, 1 if even else None)
    e(("vadd_imm", "ADDR", 8 * loop_rounds))
    e(("loop_end",))
    e(("waitcnt", 0))
    return ins


# ---- assembly printer ----------------------------------------------------------------
def v(name):
    return "v%d" % REG[name]

def vp(name):
    return "v[%d:%d]" % (REG[name + "l"], REG[name + "h"])

def to_asm(ins):
    out = []
    for t in ins:
        op = t[0]
        if op == "waitcnt":
            out.append("s_waitcnt lgkmcnt(%d)" % t[1])
        elif op == "nop":
            out.append("s_nop 0")
        elif op == "ds_read2_b64":  # two u64 at ADDR + 8*t[3] and ADDR + 8*(t[3]+1) into 4 consecutive VGPRs
            out.append("ds_read2_b64 v[%d:%d], %s offset0:%d offset1:%d" %
                       (REG[t[1] + "l"], REG[t[1] + "l"] + 3, v(t[2]), t[3], t[3] + 1))
        elif op == "alignbit":
            out.append("v_alignbit_b32 %s, %s, %s, %s" % (v(t[1]), v(t[2]), v(t[3]), v(t[4])))
        elif op == "xor":
            out.append("v_xor_b32 %s, %s, %s" % (v(t[1]), v(t[2]), v(t[3])))
        elif op == "and":
            out.append("v_and_b32 %s, %s, %s" % (v(t[1]), v(t[2]), v(t[3])))
        elif op == "bfi":
            out.append("v_bfi_b32 %s, %s, %s, %s" % (v(t[1]), v(t[2]), v(t[3]), v(t[4])))
        elif op == "bitop3":  # t[5] = truth table f(0xF0, 0xCC, 0xAA) over (src0, src1, src2)
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0x%x" % (v(t[1]), v(t[2]), v(t[3]), v(t[4]), t[5]))
        elif op == "add64":
            out.append("v_lshl_add_u64 %s, %s, 0, %s" % (vp(t[1]), vp(t[2]), vp(t[3])))
        elif op == "add_co_dpp":
            out.append("v_add_co_u32_dpp %s, vcc, %s, %s row_half_mirror row_mask:0xf bank_mask:0x%x" %
                       (v(t[1]), v(t[2]), v(t[3]), t[4]))
        elif op == "addc_co_dpp":
            out.append("v_addc_co_u32_dpp %s, vcc, %s, %s, vcc row_half_mirror row_mask:0xf bank_mask:0x%x" %
                       (v(t[1]), v(t[2]), v(t[3]), t[4]))
        elif op == "loop_begin":   # %8 = an SGPR the compiler picks (loop counter)
            out.append("s_mov_b32 %%8, %d" % t[1])
            out.append("1:")
        elif op == "vadd_imm":
            out.append("v_add_u32 %s, %d, %s" % (v(t[1]), t[2], v(t[1])))
        elif op == "loop_end":
            out.append("s_sub_u32 %8, %8, 1")
            out.append("s_cmp_lg_u32 %8, 0")
            out.append("s_cbranch_scc1 1b")
        else:
            raise ValueError(op)
    return out


LOOP_ROUNDS = 0  # rounds per hardware-loop iteration in the shipped block (0 = fully unrolled; measured fastest: 30.2 ms vs 30.6-31.9 ms for 40/16/8-round loops on C2)


def write_inc(path, loop_rounds=LOOP_ROUNDS):
    ins = build(loop_rounds=loop_rounds)
    body = to_asm(ins)
    n_valu = sum(1 for t in ins if t[0] not in ("waitcnt", "ds_read2_b64", "nop"))
    lines = ["// GENERATED by tools/gen_pair_rounds.py -- do not edit.",
             "// One 128-byte block = 80 SHA-512 rounds on lane pairs; %d VALU + %d LDS reads." %
             (n_valu, sum(1 for t in ins if t[0] == "ds_read2_b64")),
             "// Physical registers v%d..v%d (clobbered).  Operands: %%0..%%3 = the four chaining words of this lane, 64-bit," % (BASE, LAST),
             "// updated in place (chaining value + working variables: the feed-forward is done here), %4 = scratch",
             "// SGPR, %5..%7 = per-lane rotate amounts, %8 = role mask (B: ~0, A: 0), %9 = LDS byte address of the",
             "// stream's K+W row.",
             "#define SNAPHASH_PAIR_FIRST_VGPR %d" % BASE,
             "#define SNAPHASH_PAIR_LAST_VGPR %d" % LAST,
             "#define SNAPHASH_PAIR_ROUNDS_ASM \\"]
    if ALIGN in ("p2", "both", "w4", "w8", "w16"):
        lines.append('    ".p2align %s\\n" \\' % os.environ.get("SNAPHASH_PAIR_P2", "3"))
    for k in range(4):      # working variables <- chaining value: 4 x v_mov_b64 (4 bytes each)
        lines.append('    "v_mov_b64 %s, %%%d\\n" \\' % (vp("R%d" % k), k))
    for k, name in enumerate(IN_ORDER[8:]):  # 5 x v_mov_b32: with the s_waitcnt that follows, 40 bytes
        lines.append('    "v_mov_b32 %s, %%%d\\n" \\' % (v(name), 5 + k))
    for s_ in body:
        lines.append('    "%s\\n" \\' % s_)
    assert build().__len__() and 80 % 4 == 0  # after 80 rounds the register roles are back where they started
    for k in range(4):      # feed-forward: out = working + chaining value
        lines.append('    "v_lshl_add_u64 %%%d, %s, 0, %%%d\\n" \\' % (k, vp("R%d" % k), k))
    lines.append('    ""')
    clob = ", ".join('"v%d"' % n for n in range(BASE, LAST + 1))
    lines.append("#define SNAPHASH_PAIR_CLOBBERS %s, \"vcc\", \"scc\"" % clob)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(body)


# ---- lane-level simulator ----------------------------------------------------------------
def partner_index():
    lane = np.arange(64)
    return (lane & ~7) | (7 - (lane & 7))

def bank_enable(mask):
    lane = np.arange(64)
    return ((mask >> ((lane & 15) >> 2)) & 1).astype(bool)

def simulate(ins, regs, lds):
    """regs: dict name -> np.uint32[64] (keys of REG); lds: np.uint8 array (byte addressed).
    Executes the list in order (the hardware issues one wave in order, so program order is
    the semantics; s_waitcnt only has to be placed before the first use, which is asserted)."""
    P = partner_index()
    pending = []  # outstanding ds_read destinations, in issue order
    vcc = np.zeros(64, dtype=np.uint64)
    def g(n):
        for p in pending:
            assert n not in p, "read of %s before its ds_read was waited for" % n
        return regs[n]
    def g64(pair):
        return g(pair + "l").astype(np.uint64) | (g(pair + "h").astype(np.uint64) << np.uint64(32))
    # expand hardware loops into the dynamic instruction sequence
    seq, i = [], 0
    while i < len(ins):
        if ins[i][0] == "loop_begin":
            j = i + 1
            while ins[j][0] != "loop_end":
                j += 1
            seq += ins[i + 1:j] * ins[i][1]
            i = j + 1
        else:
            seq.append(ins[i])
            i += 1
    for t in seq:
        op = t[0]
        if op == "nop":
            continue
        if op == "waitcnt":
            while len(pending) > t[1]:
                pending.pop(0)
        elif op == "vadd_imm":
            regs[t[1]] = (g(t[1]).astype(np.uint64) + np.uint64(t[2])).astype(np.uint32)
        elif op == "ds_read2_b64":
            first = REG[t[1] + "l"]
            names = [n for n, r in sorted(REG.items(), key=lambda kv: kv[1]) if first <= r < first + 4]
            assert len(names) == 4 and first % 2 == 0
            addr = g(t[2]).astype(np.int64) + 8 * t[3]
            vals = [np.zeros(64, dtype=np.uint32) for _ in range(4)]
            for l in range(64):
                b = bytes(lds[addr[l]:addr[l] + 16])
                for q in range(4):
                    vals[q][l] = int.from_bytes(b[4 * q:4 * q + 4], "little")
            for n, val in zip(names, vals):
                regs[n] = val
            pending.append(tuple(names))
        elif op == "alignbit":
            hi, lo, sh = g(t[2]).astype(np.uint64), g(t[3]).astype(np.uint64), (g(t[4]) & 31).astype(np.uint64)
            regs[t[1]] = ((((hi << np.uint64(32)) | lo) >> sh) & np.uint64(M32)).astype(np.uint32)
        elif op == "xor":
            regs[t[1]] = g(t[2]) ^ g(t[3])
        elif op == "and":
            regs[t[1]] = g(t[2]) & g(t[3])
        elif op == "bfi":
            m = g(t[2])
            regs[t[1]] = (m & g(t[3])) | (~m & g(t[4]))
        elif op == "bitop3":
            a_, b_, c_ = g(t[2]), g(t[3]), g(t[4])
            r = np.zeros(64, dtype=np.uint32)
            for m in range(8):
                if t[5] & (1 << m):
                    r |= (a_ if m & 4 else ~a_) & (b_ if m & 2 else ~b_) & (c_ if m & 1 else ~c_)
            regs[t[1]] = r
        elif op == "add64":
            with np.errstate(over="ignore"):
                r = g64(t[2]) + g64(t[3])
            regs[t[1] + "l"] = (r & np.uint64(M32)).astype(np.uint32)
            regs[t[1] + "h"] = (r >> np.uint64(32)).astype(np.uint32)
        elif op in ("add_co_dpp", "addc_co_dpp"):
            en = bank_enable(t[4])
            s = g(t[2])[P].astype(np.uint64) + g(t[3]).astype(np.uint64)
            if op == "addc_co_dpp":
                s = s + vcc
            old = regs[t[1]]
            regs[t[1]] = np.where(en, (s & np.uint64(M32)).astype(np.uint32), old)
            vcc = np.where(en, s >> np.uint64(32), vcc)   # only enabled lanes are relied upon
        else:
            raise ValueError(op)
    assert not pending, "block ends with LDS reads in flight"
    return regs


def lane_roles():
    """-> (role_is_B[64], local_stream[64]) of the round wave's lane layout."""
    lane = np.arange(64)
    j = lane & 7
    is_b = j >= 4
    stream = 4 * (lane >> 3) + np.where(is_b, 7 - j, j)
    return is_b, stream


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "snappy_amd", "csrc", "pair_rounds.inc")
    loop = LOOP_ROUNDS
    if "--loop" in sys.argv:  # experiments only: the committed .inc is LOOP_ROUNDS
        loop = int(sys.argv[sys.argv.index("--loop") + 1])
    n = write_inc(out, loop)
    print("wrote %s: %d instructions per block (%.2f per round)" % (out, n, n / 80.0))
