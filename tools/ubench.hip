// ubench.hip -- gfx950 instruction-issue microbenchmarks behind the SHA-512 kernel
// design (DESIGN.md "VALU budget").  For each candidate instruction: cycles per
// wave-instruction seen by ONE wave (s_memtime around an unrolled loop), at 1, 2
// and 4 waves per SIMD, independent and dependent chains.
//   hipcc --offload-arch=gfx950 -O2 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 8192;

// 16 instructions per loop body. IND: 8 independent destinations; DEP: one chain.
#define BODY16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

#define DEFINE_KERNEL(NAME, SETUP, INSTR_IND, INSTR_DEP)                                                    \
    template <bool DEP>                                                                                      \
    __global__ __launch_bounds__(1024) void NAME(uint64_t* cycles, uint32_t* sink)                            \
    {                                                                                                        \
        uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 7, r4 = r0 * 5, r5 = r0 | 9,      \
                 r6 = r0 - 3, r7 = r0 * 11;                                                                  \
        uint32_t s0 = blockIdx.x + 1, s1 = r0 * 7 + 3;                                                       \
        uint64_t q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7, p0 = s1;            \
        SETUP                                                                                                \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                          \
        for (int it = 0; it < kIters; ++it) {                                                                \
            if (DEP) { asm volatile(BODY16(INSTR_DEP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(s0), "v"(s1), "v"(p0) : "vcc", "scc"); } \
            else { asm volatile(BODY16(INSTR_IND) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(s0), "v"(s1), "v"(p0) : "vcc", "scc"); } \
        }                                                                                                    \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                          \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;         \
        uint32_t acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7); \
        if (acc == 0x12345678) sink[0] = acc;                                                                \
    }

// operand numbering: %0..%7 = r0..r7, %8..%15 = q0..q7, %16 = s0, %17 = s1, %18 = p0
#define ALIGNBIT_I(n) "v_alignbit_b32 %" #n ", %" #n ", %17, 7\n"
#define ALIGNBIT_D(n) "v_alignbit_b32 %0, %0, %17, 7\n"
DEFINE_KERNEL(k_alignbit, , ALIGNBIT_I, ALIGNBIT_D)

#define XOR_I(n) "v_xor_b32 %" #n ", %" #n ", %17\n"
#define XOR_D(n) "v_xor_b32 %0, %0, %17\n"
DEFINE_KERNEL(k_xor, , XOR_I, XOR_D)

#define BFI_I(n) "v_bfi_b32 %" #n ", %" #n ", %16, %17\n"
#define BFI_D(n) "v_bfi_b32 %0, %0, %16, %17\n"
DEFINE_KERNEL(k_bfi, , BFI_I, BFI_D)

#define PERM_I(n) "v_perm_b32 %" #n ", %" #n ", %16, %17\n"
#define PERM_D(n) "v_perm_b32 %0, %0, %16, %17\n"
DEFINE_KERNEL(k_perm, , PERM_I, PERM_D)

#define ADD64_I(n) "v_lshl_add_u64 %" #n "+8, %" #n "+8, 0, %18\n"
#define A64I(n, m) "v_lshl_add_u64 %" #m ", %" #m ", 0, %18\n"
#define ADD64_IND(n) A64I_##n
#define A64I_0 A64I(0, 8)
#define A64I_1 A64I(1, 9)
#define A64I_2 A64I(2, 10)
#define A64I_3 A64I(3, 11)
#define A64I_4 A64I(4, 12)
#define A64I_5 A64I(5, 13)
#define A64I_6 A64I(6, 14)
#define A64I_7 A64I(7, 15)
#define ADD64_D(n) "v_lshl_add_u64 %8, %8, 0, %18\n"
DEFINE_KERNEL(k_add64, , ADD64_IND, ADD64_D)

#define SHR64I(n, m) "v_lshrrev_b64 %" #m ", 7, %" #m "\n"
#define SHR64_IND(n) SHR64I_##n
#define SHR64I_0 SHR64I(0, 8)
#define SHR64I_1 SHR64I(1, 9)
#define SHR64I_2 SHR64I(2, 10)
#define SHR64I_3 SHR64I(3, 11)
#define SHR64I_4 SHR64I(4, 12)
#define SHR64I_5 SHR64I(5, 13)
#define SHR64I_6 SHR64I(6, 14)
#define SHR64I_7 SHR64I(7, 15)
#define SHR64_D(n) "v_lshrrev_b64 %8, 7, %8\n"
DEFINE_KERNEL(k_shr64, , SHR64_IND, SHR64_D)

// 64-bit add as the classic carry pair (2 instructions per body slot -> 32 per loop)
#define ADDC_I(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %16\nv_addc_co_u32 %" #n ", vcc, %" #n ", %17, vcc\n"
#define ADDC_D(n) "v_add_co_u32 %0, vcc, %0, %16\nv_addc_co_u32 %0, vcc, %0, %17, vcc\n"
DEFINE_KERNEL(k_addc_pair, , ADDC_I, ADDC_D)

#define DPPMOV_I(n) "v_mov_b32_dpp %" #n ", %17 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define DPPMOV_D(n) "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
DEFINE_KERNEL(k_dppmov, , DPPMOV_I, DPPMOV_D)

#define DPPADD_I(n) "v_add_co_u32_dpp %" #n ", vcc, %17, %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define DPPADD_D(n) "v_add_co_u32_dpp %0, vcc, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
DEFINE_KERNEL(k_dppadd, , DPPADD_I, DPPADD_D)

#define DPPADDM_I(n) "v_add_co_u32_dpp %" #n ", vcc, %17, %" #n " row_half_mirror row_mask:0xf bank_mask:0x5\nv_addc_co_u32_dpp %" #n ", vcc, %17, %" #n ", vcc row_half_mirror row_mask:0xf bank_mask:0x5\n"
#define DPPADDM_D(n) "v_add_co_u32_dpp %0, vcc, %17, %0 row_half_mirror row_mask:0xf bank_mask:0x5\nv_addc_co_u32_dpp %0, vcc, %17, %0, vcc row_half_mirror row_mask:0xf bank_mask:0x5\n"
DEFINE_KERNEL(k_dppadd_pair_masked, , DPPADDM_I, DPPADDM_D)

#define DPPADDU_I(n) "v_add_co_u32_dpp %" #n ", vcc, %17, %" #n " row_half_mirror row_mask:0xf bank_mask:0xf\nv_addc_co_u32_dpp %" #n ", vcc, %17, %" #n ", vcc row_half_mirror row_mask:0xf bank_mask:0xf\n"
#define DPPADDU_D(n) "v_add_co_u32_dpp %0, vcc, %17, %0 row_half_mirror row_mask:0xf bank_mask:0xf\nv_addc_co_u32_dpp %0, vcc, %17, %0, vcc row_half_mirror row_mask:0xf bank_mask:0xf\n"
DEFINE_KERNEL(k_dppadd_pair_unmasked, , DPPADDU_I, DPPADDU_D)

#define DPPXOR_I(n) "v_xor_b32_dpp %" #n ", %17, %" #n " quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n"
#define DPPXOR_D(n) "v_xor_b32_dpp %0, %17, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n"
DEFINE_KERNEL(k_dppxor_masked, , DPPXOR_I, DPPXOR_D)

#define SWAP32_I(n) "v_permlane32_swap_b32 %" #n ", %17\n"
#define SWAP32_D(n) "v_permlane32_swap_b32 %0, %0\n"
DEFINE_KERNEL(k_permlane32_swap, , SWAP32_I, SWAP32_D)

#define HMIR_I(n) "v_mov_b32_dpp %" #n ", %17 row_half_mirror row_mask:0xf bank_mask:0xf\n"
#define HMIR_D(n) "v_mov_b32_dpp %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
DEFINE_KERNEL(k_dpp_half_mirror, , HMIR_I, HMIR_D)

#define CNDMASK_I(n) "v_cndmask_b32 %" #n ", %" #n ", %17, vcc\n"
#define CNDMASK_D(n) "v_cndmask_b32 %0, %0, %17, vcc\n"
DEFINE_KERNEL(k_cndmask, , CNDMASK_I, CNDMASK_D)

// SALU op between VALU ops: does a scalar instruction cost the wave an issue slot?
#define VS_I(n) "v_xor_b32 %" #n ", %" #n ", %17\ns_lshl_b64 vcc, vcc, 1\n"
#define VS_D(n) "v_xor_b32 %0, %0, %17\ns_lshl_b64 vcc, vcc, 1\n"
DEFINE_KERNEL(k_valu_plus_salu, , VS_I, VS_D)

// Mixes (round 3): does a full-rate instruction keep its rate between half-rate ones?  The WIDE kernel's block is
// 2 352 half-rate (v_alignbit, v_lshl_add_u64, v_perm) + ~1 020 full-rate (v_bitop3, shifts) instructions.
#define MIXAB_I(n) "v_alignbit_b32 %" #n ", %" #n ", %17, 7\nv_bitop3_b32 %" #n ", %" #n ", %16, %17 bitop3:0x96\n"
#define MIXAB_D(n) "v_alignbit_b32 %0, %0, %17, 7\nv_bitop3_b32 %0, %0, %16, %17 bitop3:0x96\n"
DEFINE_KERNEL(k_mix_alignbit_bitop3, , MIXAB_I, MIXAB_D)
// three half-rate, one full-rate: the ratio of a SHA-512 round (19 : 8)
#define MIX31_I(n) "v_alignbit_b32 %" #n ", %" #n ", %17, 7\nv_alignbit_b32 %" #n ", %" #n ", %16, 9\nv_bitop3_b32 %" #n ", %" #n ", %16, %17 bitop3:0x96\nv_alignbit_b32 %" #n ", %" #n ", %17, 3\n"
#define MIX31_D(n) "v_alignbit_b32 %0, %0, %17, 7\nv_alignbit_b32 %0, %0, %16, 9\nv_bitop3_b32 %0, %0, %16, %17 bitop3:0x96\nv_alignbit_b32 %0, %0, %17, 3\n"
DEFINE_KERNEL(k_mix_3alignbit_1bitop3, , MIX31_I, MIX31_D)
#define MIXXX_I(n) "v_bitop3_b32 %" #n ", %" #n ", %16, %17 bitop3:0x96\nv_xor_b32 %" #n ", %" #n ", %17\n"
#define MIXXX_D(n) "v_bitop3_b32 %0, %0, %16, %17 bitop3:0x96\nv_xor_b32 %0, %0, %17\n"
DEFINE_KERNEL(k_mix_bitop3_xor, , MIXXX_I, MIXXX_D)

// LDS read beside VALU: one ds_read_b64 per 4 VALU (the pair kernel's K+W fetch)
template <bool DEP>
__global__ __launch_bounds__(1024) void k_lds_mix(uint64_t* cycles, uint32_t* sink)
{
    __shared__ uint64_t tab[1024];
    tab[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t r0 = threadIdx.x, s1 = r0 * 7 + 3;
    uint32_t addr = (threadIdx.x & 63) * 8;
    uint64_t acc = 0, v;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("ds_read_b64 %0, %1\n" : "=v"(v) : "v"(addr));
            asm volatile("v_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\n" : "+v"(r0) : "v"(s1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += v;
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if ((uint32_t)acc + r0 == 0x12345678) sink[0] = r0;
}

// Cross-wave hand-off inside a workgroup: every wave writes a value to LDS, s_barrier, reads its
// neighbour wave's value (dependent chain: the value read feeds the next write).  Cycles per
// round trip = the price of splitting one stream's round across waves.
template <bool DEP>
__global__ __launch_bounds__(1024) void k_lds_barrier_pingpong(uint64_t* cycles, uint32_t* sink)
{
    __shared__ uint32_t box[1024];
    const uint32_t t = threadIdx.x, peer = (t + 64u) % blockDim.x;
    uint32_t v = t;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 2048; ++it) {
        box[t] = v;
        __syncthreads();
        v = box[peer] + 1u;
        if (DEP) __syncthreads(); // second barrier: the slot may be rewritten only after everyone has read
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((t & 63) == 0) cycles[(blockIdx.x * blockDim.x + t) >> 6] = t1 - t0;
    if (v == 0x12345678) sink[0] = v;
}


// ---- round 2 additions -------------------------------------------------------------------
// CDNA4's three-input truth-table op (xor3, Ch, Maj in one instruction)
#define BITOP3_I(n) "v_bitop3_b32 %" #n ", %" #n ", %16, %17 bitop3:0x96\n"
#define BITOP3_D(n) "v_bitop3_b32 %0, %0, %16, %17 bitop3:0x96\n"
DEFINE_KERNEL(k_bitop3, , BITOP3_I, BITOP3_D)

// the SAME operation in the 4-byte (VOP2) and the 8-byte (VOP3) encoding: is the half rate of
// alignbit/bfi/perm/lshl_add a property of the encoding width or of the operation?
#define XOR64_I(n) "v_xor_b32_e64 %" #n ", %" #n ", %17\n"
#define XOR64_D(n) "v_xor_b32_e64 %0, %0, %17\n"
DEFINE_KERNEL(k_xor_e64, , XOR64_I, XOR64_D)

#define ADDU32_I(n) "v_add_u32_e32 %" #n ", %" #n ", %17\n"
#define ADDU32_D(n) "v_add_u32_e32 %0, %0, %17\n"
DEFINE_KERNEL(k_add_u32_e32, , ADDU32_I, ADDU32_D)

#define ADDU64E_I(n) "v_add_u32_e64 %" #n ", %" #n ", %17\n"
#define ADDU64E_D(n) "v_add_u32_e64 %0, %0, %17\n"
DEFINE_KERNEL(k_add_u32_e64, , ADDU64E_I, ADDU64E_D)

#define FMA_I(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define FMA_D(n) "v_fma_f32 %0, %0, %16, %17\n"
DEFINE_KERNEL(k_fma_f32, , FMA_I, FMA_D)

#define ADD3_I(n) "v_add3_u32 %" #n ", %" #n ", %16, %17\n"
#define ADD3_D(n) "v_add3_u32 %0, %0, %16, %17\n"
DEFINE_KERNEL(k_add3_u32, , ADD3_I, ADD3_D)

#define MOV_I(n) "v_mov_b32_e32 %" #n ", %17\n"
#define MOV_D(n) "v_mov_b32_e32 %0, %0\n"
DEFINE_KERNEL(k_mov_e32, , MOV_I, MOV_D)

#define ROR8_I(n) "v_mov_b32_dpp %" #n ", %17 row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define ROR8_D(n) "v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
DEFINE_KERNEL(k_dpp_row_ror8, , ROR8_I, ROR8_D)

#define SHR8_I(n) "v_add_u32_dpp %" #n ", %17, %" #n " row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define SHR8_D(n) "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
DEFINE_KERNEL(k_dppadd_row_shr8, , SHR8_I, SHR8_D)

#define MAD64I(n, m) "v_mad_u64_u32 %" #m ", vcc, %" #n ", %16, %" #m "\n"
#define MAD64_IND(n) MAD64I_##n
#define MAD64I_0 MAD64I(0, 8)
#define MAD64I_1 MAD64I(1, 9)
#define MAD64I_2 MAD64I(2, 10)
#define MAD64I_3 MAD64I(3, 11)
#define MAD64I_4 MAD64I(4, 12)
#define MAD64I_5 MAD64I(5, 13)
#define MAD64I_6 MAD64I(6, 14)
#define MAD64I_7 MAD64I(7, 15)
#define MAD64_D(n) "v_mad_u64_u32 %8, vcc, %0, %16, %8\n"
DEFINE_KERNEL(k_mad_u64_u32, , MAD64_IND, MAD64_D)

// Two waves on one SIMD: waves 0-3 of a workgroup land on the four SIMDs, wave 4+ share them.
// Every wave runs the PAIR round wave's instruction mix (8 alignbit/bfi, 4 bitop3, 3 lshl_add_u64,
// 4 DPP adds per "round"); waves 0-3 at s_setprio 3 (as the round waves are), the rest at 0 and
// `duty` of 4 iterations busy.  Reported: cycles per "round" of each wave -- what a round wave
// loses when a helper wave shares its SIMD.
__global__ __launch_bounds__(1024) void k_share(uint64_t* cycles, uint32_t* sink, int iters, int duty)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 7, r4 = r0 * 5, r5 = r0 | 9, r6 = r0 - 3, r7 = r0 * 11;
    uint32_t s0 = 13, s1 = r0 * 7 + 3;
    uint64_t q0 = r0, q1 = r1, q2 = r2;
    if (wave < 4) __builtin_amdgcn_s_setprio(3);
    const int n = wave < 4 ? iters : (iters * duty) / 4;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        asm volatile(
            "v_alignbit_b32 %0, %1, %0, %12\n" "v_alignbit_b32 %1, %0, %1, %12\n"
            "v_alignbit_b32 %2, %1, %0, %11\n" "v_alignbit_b32 %3, %0, %1, %11\n"
            "v_bitop3_b32 %4, %0, %2, %5 bitop3:0x78\n" "v_bitop3_b32 %5, %1, %3, %4 bitop3:0x78\n"
            "v_bitop3_b32 %0, %0, %2, %6 bitop3:0x96\n" "v_bitop3_b32 %1, %1, %3, %7 bitop3:0x96\n"
            "v_bfi_b32 %6, %4, %2, %0\n" "v_bfi_b32 %7, %5, %3, %1\n"
            "v_lshl_add_u64 %8, %8, 0, %9\n"
            "v_alignbit_b32 %2, %1, %0, %12\n" "v_alignbit_b32 %3, %0, %1, %12\n"
            "v_lshl_add_u64 %9, %9, 0, %10\n" "v_lshl_add_u64 %10, %10, 0, %8\n"
            "v_add_co_u32_dpp %4, vcc, %6, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n"
            "v_addc_co_u32_dpp %5, vcc, %7, %5, vcc row_half_mirror row_mask:0xf bank_mask:0x5\n"
            "v_add_co_u32_dpp %4, vcc, %2, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n"
            "v_addc_co_u32_dpp %5, vcc, %3, %5, vcc row_half_mirror row_mask:0xf bank_mask:0xa\n"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(q0), "+v"(q1), "+v"(q2)
            : "v"(s0), "v"(s1) : "vcc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
    uint32_t acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)(q0 ^ q1 ^ q2);
    if (acc == 0x12345678) sink[0] = acc;
}

static void run_share(int nwaves, int duty, uint64_t* d_cycles, uint32_t* d_sink)
{
    const int iters = 20000, blocks = 256;
    CHECK(hipMemset(d_cycles, 0, blocks * 16 * 8));
    hipLaunchKernelGGL(k_share, dim3(blocks), dim3(64 * nwaves), 0, 0, d_cycles, d_sink, iters, duty);
    hipLaunchKernelGGL(k_share, dim3(blocks), dim3(64 * nwaves), 0, 0, d_cycles, d_sink, iters, duty);
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> h(blocks * 16);
    CHECK(hipMemcpy(h.data(), d_cycles, blocks * 16 * 8, hipMemcpyDeviceToHost));
    printf("k_share waves/WG=%d helper duty=%d/4: cycles per 19-instruction round, median over 256 CUs:", nwaves, duty);
    for (int w = 0; w < nwaves; ++w) {
        std::vector<uint64_t> c;
        for (int b = 0; b < blocks; ++b) c.push_back(h[b * 16 + w]);
        std::sort(c.begin(), c.end());
        const int n = w < 4 ? iters : (iters * duty) / 4;
        printf("  w%d=%.1f", w, (double)c[blocks / 2] / n);
    }
    printf("\n");
}

template <typename K>
static void run(const char* name, K kern, int instr_per_loop, int waves_per_simd, uint64_t* d_cycles, uint32_t* d_sink)
{
    // one workgroup per CU, 4*waves_per_simd waves each; waves_per_simd < 0 means -waves_per_simd
    // waves per CU in total (1 or 2: fewer waves than SIMDs)
    const int threads = waves_per_simd > 0 ? 256 * waves_per_simd : 64 * -waves_per_simd;
    const int blocks = 256;
    const int nw = blocks * threads / 64;
    CHECK(hipMemset(d_cycles, 0, nw * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink); // warm-up
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint64_t> h(nw);
    CHECK(hipMemcpy(h.data(), d_cycles, nw * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double per = (double)h[nw / 2] / ((double)kIters * instr_per_loop);
    // s_memtime ticks at 100 MHz on gfx9 ("shader clock" per the guide: treat as-is and also give wall)
    const double wave_instr = (double)nw * kIters * instr_per_loop;
    printf("%-22s waves/SIMD=%d  memtime/instr=%7.3f  wall=%8.3f ms  chip Ginstr/s=%8.2f\n", name, waves_per_simd, per, ms,
           wave_instr / (ms * 1e-3) / 1e9);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    printf("ubench start\n");
    uint64_t* d_cycles;
    uint32_t* d_sink;
    CHECK(hipMalloc(&d_cycles, 256 * 16 * 8 * 2));
    CHECK(hipMalloc(&d_sink, 64));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : {-1, 1, 2, 4}) {
#define RUN(k, n) run(#k " ind", k<false>, n, w, d_cycles, d_sink); run(#k " dep", k<true>, n, w, d_cycles, d_sink)
        RUN(k_alignbit, 16);
        RUN(k_xor, 16);
        RUN(k_bfi, 16);
        RUN(k_perm, 16);
        RUN(k_add64, 16);
        RUN(k_shr64, 16);
        RUN(k_addc_pair, 32);
        RUN(k_dppmov, 16);
        RUN(k_dppadd, 16);
        RUN(k_dpp_half_mirror, 16);
        RUN(k_dppadd_pair_masked, 32);
        RUN(k_dppadd_pair_unmasked, 32);
        RUN(k_dppxor_masked, 16);
        RUN(k_permlane32_swap, 16);
        RUN(k_cndmask, 16);
        RUN(k_bitop3, 16);
        RUN(k_xor_e64, 16);
        RUN(k_add_u32_e32, 16);
        RUN(k_add_u32_e64, 16);
        RUN(k_fma_f32, 16);
        RUN(k_add3_u32, 16);
        RUN(k_mov_e32, 16);
        RUN(k_dpp_row_ror8, 16);
        RUN(k_dppadd_row_shr8, 16);
        RUN(k_mad_u64_u32, 16);
        RUN(k_valu_plus_salu, 32);
        RUN(k_mix_alignbit_bitop3, 32);
        RUN(k_mix_3alignbit_1bitop3, 64);
        RUN(k_mix_bitop3_xor, 32);
        run("k_lds_mix(1ds+4valu)", k_lds_mix<false>, 20, w, d_cycles, d_sink);
        // per-iteration cost: kIters is not used by this kernel (2048 iterations): scale = 2048/kIters per "instr"
        run("lds+barrier hop x1 (cycles*4/iter)", k_lds_barrier_pingpong<false>, 1, w, d_cycles, d_sink);
        run("lds+2barriers hop (cycles*4/iter)", k_lds_barrier_pingpong<true>, 1, w, d_cycles, d_sink);
        printf("\n");
    }
    for (int nw : {4, 5, 6, 8})
        for (int duty : {4, 2}) {
            if (nw == 4 && duty != 4) continue;
            run_share(nw, duty, d_cycles, d_sink);
        }
    return 0;
}
