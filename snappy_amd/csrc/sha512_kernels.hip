// sha512_kernels.hip -- multi-buffer SHA-512 for gfx950 (MI355X, CDNA4).
//
// The GPU side of helpers.Sha512sum (reference helpers/helpers.go:187-201)
// batched over the files of one writeHashes pass (snappy/build.go:228-259).
// A file's blocks are strictly sequential (Merkle-Damgard chaining), so the
// parallel axis is the file list: every kernel here advances many independent
// streams in lockstep.  Three hashing kernels for three regimes:
//   sha512_wide_kernel          one lane per stream; saturates the VALUs (many streams)
//   sha512_split_kernel<false>  rounds on one wave, message schedule on helper waves
//   sha512_split_kernel<true>   the same with every stream on a lane pair (generated
//                               assembly, pair_rounds.inc): fewest instructions on the
//                               wave that carries the chaining value (few long streams)
// plus ranges_equal_kernel (helpers.FilesAreEqual's byte compare, HBM-bound) and the
// synthetic-content generator used by tests and bench.
//
// Data layout in HBM: file bytes are contiguous, 16-byte aligned, described by
// one 32-byte Job per stream segment.  A wave fetches 128-byte blocks with
// full-line coalesced dwordx4 loads (8 lanes per stream, 8 streams per load
// instruction), stages them in a wave-private LDS tile (144-byte row stride to
// spread ds_read_b128 over the banks) and each lane then reads its own block.
// One block is prefetched into registers while the previous one is hashed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pair_rounds.inc"
#if defined(SNAPHASH_WITH_QUAD) // the measured-negative four-lane variant: `make QUAD=1`
#include "quad_rounds.inc"
#endif
#include "sha512_core.h"
#include "sha512_kernels.h"

namespace snaphash {

namespace {

constexpr int kTileRow = 9; // uint4 per stream row: 8 data + 1 pad (144 B)

// Round constants in the constant address space: uniform indexing turns into
// s_load_dwordx16, the SGPR pairs feed v_lshl_add_u64 directly.
__constant__ uint64_t d_K512[80] = {SNAPHASH_K512_LIST};

// Loads through an explicit global-address-space pointer (global_load_dwordx4,
// not flat_load: the Job carries the address as an integer).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;
__device__ __forceinline__ uint4 load_u4(const uint8_t* p)
{
    const u32x4 v = *(gptr_u4)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    return ((uint64_t)shfl_u32((uint32_t)(v >> 32), src) << 32) | shfl_u32((uint32_t)v, src);
}

__device__ __forceinline__ void store_digest_be(uint8_t* out, const uint64_t H[8])
{
    uint4* o = reinterpret_cast<uint4*>(out);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t x = H[2 * k], y = H[2 * k + 1];
        uint4 v;
        v.x = __builtin_bswap32((uint32_t)(x >> 32));
        v.y = __builtin_bswap32((uint32_t)x);
        v.z = __builtin_bswap32((uint32_t)(y >> 32));
        v.w = __builtin_bswap32((uint32_t)y);
        o[k] = v;
    }
}

// ---------------------------------------------------------------------------
// WIDE kernel: one lane per stream, 64 streams per wave, one wave per
// workgroup.  The efficient form when there are enough streams to fill the
// chip: every VALU instruction advances 64 streams; saturates the VALUs at
// ~1.05 TB/s from 65 536 streams on (profiles/r01_regime_sweep.txt).
// ---------------------------------------------------------------------------
#if defined(SNAPHASH_WIDE_WAVES) // experiment: ask the register allocator for that many waves per SIMD
#define SNAPHASH_WIDE_ATTR __attribute__((amdgpu_waves_per_eu(SNAPHASH_WIDE_WAVES, SNAPHASH_WIDE_WAVES)))
#else
#define SNAPHASH_WIDE_ATTR
#endif
__global__ __launch_bounds__(64) SNAPHASH_WIDE_ATTR void sha512_wide_kernel(const Job* __restrict__ jobs, uint32_t njobs,
                                                         uint64_t* __restrict__ state,
                                                         uint8_t* __restrict__ digests)
{
    __shared__ uint4 tile[64 * kTileRow];
    const uint32_t lane = threadIdx.x;
    const uint32_t slot = blockIdx.x * 64u + lane;
    const bool have = slot < njobs;

    Job jb;
    if (have) {
        jb = jobs[slot];
    } else {
        jb.data = 0; jb.nbytes = 0; jb.total_prev = 0; jb.idx = 0; jb.flags = 0;
    }
    const uint64_t nbytes = jb.nbytes;
    const uint32_t nfull = (uint32_t)(nbytes >> 7);
    const uint32_t rem = (uint32_t)(nbytes & 127);
    const bool fin = (jb.flags & kJobFinal) != 0;
    const uint32_t nblk = have ? padded_blocks(nbytes, fin) : 0u;
    const uint64_t total = jb.total_prev + nbytes;

    // Cooperative loader: load instruction i covers streams 8i..8i+7, lane
    // l fetches 16-byte piece (l & 7) of stream 8i + (l >> 3).
    const uint32_t piece = lane & 7u;
    const uint8_t* tptr[8];
    uint32_t tnp[8]; // 16-byte pieces the target stream holds (last one may be partial)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = 8 * i + (int)(lane >> 3);
        const uint64_t d = shfl_u64(jb.data, t);
        const uint64_t nb = shfl_u64(nbytes, t);
        tptr[i] = reinterpret_cast<const uint8_t*>(d) + piece * 16u;
        tnp[i] = (uint32_t)((nb + 15u) >> 4);
    }

    uint64_t H[8];
    if (jb.flags & kJobFirst) {
#pragma unroll
        for (int k = 0; k < 8; ++k) H[k] = IV512[k];
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) H[k] = have ? state[(uint64_t)jb.idx * 8 + k] : 0;
    }

    uint4 pre[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pre[i] = make_uint4(0, 0, 0, 0);
        if (piece < tnp[i]) pre[i] = load_u4(tptr[i]);
    }

    for (uint32_t b = 0; __any(b < nblk); ++b) {
        __syncthreads(); // single wave: orders last iteration's tile reads before these writes
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[(8 * i + (lane >> 3)) * kTileRow + piece] = pre[i];
        __syncthreads();

        // prefetch block b+1 while block b is hashed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t p = (b + 1u) * 8u + piece;
            pre[i] = make_uint4(0, 0, 0, 0);
            if (p < tnp[i]) pre[i] = load_u4(tptr[i] + (uint64_t)(b + 1u) * 128u);
        }

        uint64_t w[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 q = tile[lane * kTileRow + k];
            w[2 * k] = be64(q.x, q.y);
            w[2 * k + 1] = be64(q.z, q.w);
        }
        if (__any(b >= nfull && b < nblk)) { // wave-uniform: only near a stream's end
            apply_padding(w, b >= nfull, b - nfull, rem, total);
        }
        compress_block(H, w, b < nblk, d_K512);
    }

    if (have) {
        if (fin) {
            store_digest_be(digests + (uint64_t)jb.idx * 64, H);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) state[(uint64_t)jb.idx * 8 + k] = H[k];
        }
    }
}

#if defined(SNAPHASH_EXPERIMENT_WIDE_DIRECT)
// ---------------------------------------------------------------------------
// WIDE, direct form: as above without the LDS tile.  Every lane fetches ITS OWN 128-byte block with eight
// global_load_dwordx4 (one cache line per lane, all of it used by the eight loads back to back: the HBM traffic
// stays 1x, the address unit sees 64 lines per instruction -- ~0.5 k cycles of it per 16 k-cycle block).  No tile,
// no cooperative pointers, no register prefetch: 56 VGPRs less, no LDS, so 5 waves per SIMD instead of 3 and the
// other waves hide the load latency.  EXPERIMENT (make WIDE_DIRECT=1, then SNAPHASH_WIDE_FORM=1|2|3 at run time):
// bit-exact and no faster at 5, 6 or 8 waves per SIMD -- the saturated regime is bound by the vector unit's rate for
// this instruction mix, not by latency (profiles/r03_wide_saturated_explained.txt).
// ---------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void sha512_wide_direct_kernel(
    const Job* __restrict__ jobs, uint32_t njobs, uint64_t* __restrict__ state, uint8_t* __restrict__ digests)
{
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    const bool have = slot < njobs;
    Job jb;
    if (have) {
        jb = jobs[slot];
    } else {
        jb.data = 0; jb.nbytes = 0; jb.total_prev = 0; jb.idx = 0; jb.flags = 0;
    }
    const uint64_t nbytes = jb.nbytes;
    const uint32_t nfull = (uint32_t)(nbytes >> 7);
    const uint32_t rem = (uint32_t)(nbytes & 127);
    const bool fin = (jb.flags & kJobFinal) != 0;
    const uint32_t nblk = have ? padded_blocks(nbytes, fin) : 0u;
    const uint64_t total = jb.total_prev + nbytes;
    const uint32_t npieces = (uint32_t)((nbytes + 15u) >> 4); // 16-byte pieces the stream holds (the last may be partial: the slack behind a stream is readable)
    const uint8_t* p = reinterpret_cast<const uint8_t*>(jb.data);

    uint64_t H[8];
    if (jb.flags & kJobFirst) {
#pragma unroll
        for (int k = 0; k < 8; ++k) H[k] = IV512[k];
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) H[k] = have ? state[(uint64_t)jb.idx * 8 + k] : 0;
    }
    for (uint32_t b = 0; __any(b < nblk); ++b) {
        uint64_t w[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint4 q = make_uint4(0, 0, 0, 0);
            if (b * 8u + (uint32_t)k < npieces) q = load_u4(p + (uint64_t)b * 128u + 16u * k);
            w[2 * k] = be64(q.x, q.y);
            w[2 * k + 1] = be64(q.z, q.w);
        }
        if (__any(b >= nfull && b < nblk)) apply_padding(w, b >= nfull, b - nfull, rem, total);
        compress_block(H, w, b < nblk, d_K512);
    }
    if (have) {
        if (fin) {
            store_digest_be(digests + (uint64_t)jb.idx * 64, H);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) state[(uint64_t)jb.idx * 8 + k] = H[k];
        }
    }
}

#endif // SNAPHASH_EXPERIMENT_WIDE_DIRECT

// ---------------------------------------------------------------------------
// SPLIT kernel: for the stream-starved regime (fewer streams than the chip has
// SIMDs x 64 lanes -- BASELINE config 2 has 10 001).  There a wave sits alone on
// its SIMD and issues one VALU instruction per ~5 cycles whatever it does, so a
// stream's speed is set by the instruction count of the wave that carries its
// chaining value.  The 80 rounds need the chaining value; the message schedule
// (W[16..79], 36 % of the work) does not.  So per 64 streams:
//   wave 0      "round wave": only the 80 rounds, fed K[t]+W[t] from LDS
//   waves 1, 2  "helper waves": fetch the 128-byte blocks (coalesced, staged in
//               a wave-private LDS tile), byte-swap, pad, expand the schedule
//               and store K[t]+W[t]; they take alternate blocks (wave 1 even,
//               wave 2 odd) and have two block-times per block.
// Hand-over through a ring of three 64x80-word K+W slots in LDS (row stride 81
// words: conflict-free ds_read_b64 across streams), one workgroup barrier per
// block-time.  One workgroup per CU (140 KB of LDS), each wave on its own SIMD.
// ---------------------------------------------------------------------------
constexpr int kKwRow = 81; // u64 per stream row of a K+W slot: 80 + 1 pad

struct SplitShared {
    uint64_t kw[3][64 * kKwRow];   // 3 x 41 472 B
    uint4 tile[2][64 * kTileRow];  // one staging tile per helper wave
    uint32_t maxblk;
    uint32_t pad_[3];
    uint64_t zeros[kKwRow];        // QUAD: the a-chain lanes read their "K+W" here (they add d + 0)
};

__device__ __forceinline__ void load_kw16(uint64_t k[16], const uint64_t* __restrict__ row)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = row[i];
}

__device__ __forceinline__ void rounds16(uint64_t& a, uint64_t& b, uint64_t& c, uint64_t& d, uint64_t& e, uint64_t& f,
                                         uint64_t& g, uint64_t& h, const uint64_t k[16])
{
    SNAPHASH_ROUND(a, b, c, d, e, f, g, h, k[0]);
    SNAPHASH_ROUND(h, a, b, c, d, e, f, g, k[1]);
    SNAPHASH_ROUND(g, h, a, b, c, d, e, f, k[2]);
    SNAPHASH_ROUND(f, g, h, a, b, c, d, e, k[3]);
    SNAPHASH_ROUND(e, f, g, h, a, b, c, d, k[4]);
    SNAPHASH_ROUND(d, e, f, g, h, a, b, c, k[5]);
    SNAPHASH_ROUND(c, d, e, f, g, h, a, b, k[6]);
    SNAPHASH_ROUND(b, c, d, e, f, g, h, a, k[7]);
    SNAPHASH_ROUND(a, b, c, d, e, f, g, h, k[8]);
    SNAPHASH_ROUND(h, a, b, c, d, e, f, g, k[9]);
    SNAPHASH_ROUND(g, h, a, b, c, d, e, f, k[10]);
    SNAPHASH_ROUND(f, g, h, a, b, c, d, e, k[11]);
    SNAPHASH_ROUND(e, f, g, h, a, b, c, d, k[12]);
    SNAPHASH_ROUND(d, e, f, g, h, a, b, c, k[13]);
    SNAPHASH_ROUND(c, d, e, f, g, h, a, b, k[14]);
    SNAPHASH_ROUND(b, c, d, e, f, g, h, a, k[15]);
}

template <int T0, int T1>
__device__ __forceinline__ void schedule_span(uint64_t w[16], uint64_t* __restrict__ row)
{
#pragma unroll
    for (int t = T0; t < T1; ++t) {
        if (t >= 16) w[t & 15] += small_sigma1(w[(t + 14) & 15]) + w[(t + 9) & 15] + small_sigma0(w[(t + 1) & 15]);
#if defined(SNAPHASH_EXPERIMENT_NO_KW_STORE) // timing experiment only (wrong digests): the schedule is computed, never stored
        { uint64_t v_ = w[t & 15] + K512[t]; asm volatile("" ::"v"(v_)); (void)row; }
#else
        row[t] = w[t & 15] + K512[t];
#endif
    }
}

// PAIR variant of the round wave(s): a stream is carried by a lane pair (role A:
// e,f,g,h and T1; role B: a,b,c,d and T2) and the 80 rounds are the generated
// assembly of pair_rounds.inc (tools/gen_pair_rounds.py): 24 instead of 32 VALU
// instructions per round.  Two round waves of 32 streams each per workgroup.
__device__ __forceinline__ void pair_round_wave(SplitShared& sh, uint32_t rw, uint32_t lane, const Job* __restrict__ jobs,
                                                uint32_t njobs, uint32_t steps, uint64_t* __restrict__ state,
                                                uint8_t* __restrict__ digests)
{
    const uint32_t j = lane & 7u;
    const bool is_b = j >= 4u;
    const uint32_t sl = 32u * rw + 4u * (lane >> 3) + (is_b ? 7u - j : j); // stream within the workgroup
    const uint32_t slot = blockIdx.x * 64u + sl;
    const bool have = slot < njobs;
    Job jb;
    if (have) {
        jb = jobs[slot];
    } else {
        jb.data = 0; jb.nbytes = 0; jb.total_prev = 0; jb.idx = 0; jb.flags = 0;
    }
    const bool fin = (jb.flags & kJobFinal) != 0;
    const uint32_t nblk = have ? padded_blocks(jb.nbytes, fin) : 0u;
    const uint32_t half = is_b ? 0u : 4u; // B owns H[0..3] = a,b,c,d ; A owns H[4..7] = e,f,g,h
    uint64_t Hx[4];
    if (jb.flags & kJobFirst) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Hx[k] = is_b ? IV512[k] : IV512[4 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) Hx[k] = have ? state[(uint64_t)jb.idx * 8 + half + k] : 0;
    }
    // per-lane rotate amounts: Sigma1(e) = rotr14(e ^ rotr4 e ^ rotr27 e), Sigma0(a) = rotr28(a ^ rotr6 a ^ rotr11 a)
    const uint32_t c1 = is_b ? 6u : 4u, c2 = is_b ? 11u : 27u, c3 = is_b ? 28u : 14u;
    const uint32_t mb = is_b ? 0xffffffffu : 0u;
    uint32_t ring = 0;
    __builtin_amdgcn_s_setprio(3); // the chaining-value wave is the critical path: win every arbitration on the CU
    for (uint32_t tau = 0; tau < steps; ++tau) {
        __syncthreads();
        if (tau < 2) continue;
        const uint32_t b = tau - 2u;
        const uint32_t addr =
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(const void*)&sh.kw[ring][sl * kKwRow];
        ring = (ring == 2u) ? 0u : ring + 1u;
        uint32_t loop_counter; // scratch SGPR of the generated block
        // the chaining words are updated in place: 80 rounds + the feed-forward add, all inside the generated block
        asm volatile(SNAPHASH_PAIR_ROUNDS_ASM
                     : "+v"(Hx[0]), "+v"(Hx[1]), "+v"(Hx[2]), "+v"(Hx[3]), "=&s"(loop_counter)
                     : "v"(c1), "v"(c2), "v"(c3), "v"(mb), "v"(addr)
                     : SNAPHASH_PAIR_CLOBBERS, "memory");
        // A lane stores its result in the block-time its stream ends and rides along afterwards with a
        // chaining value nobody reads: no per-block select (8 v_cndmask at ~19 cycles each for a lone wave).
        if (b + 1u == nblk) {
            if (fin) {
                uint4* o = reinterpret_cast<uint4*>(digests + (uint64_t)jb.idx * 64 + half * 8);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    uint4 v;
                    v.x = __builtin_bswap32(hi32(Hx[2 * k]));
                    v.y = __builtin_bswap32(lo32(Hx[2 * k]));
                    v.z = __builtin_bswap32(hi32(Hx[2 * k + 1]));
                    v.w = __builtin_bswap32(lo32(Hx[2 * k + 1]));
                    o[k] = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) state[(uint64_t)jb.idx * 8 + half + k] = Hx[k];
            }
        }
    }
}

// Helper wave hk (0: even blocks, 1: odd blocks) of a 64-stream workgroup: fetch the 128-byte blocks
// (coalesced, staged in a wave-private LDS tile), byte-swap, pad, expand the message schedule and store
// K[t]+W[t] into the ring; lane = stream.  Shared by the SPLIT, PAIR and QUAD kernels.
__device__ __forceinline__ void split_helper_wave(SplitShared& sh, uint32_t hk, uint32_t lane, const Job& jb, uint32_t nblk,
                                                  uint32_t steps)
{
    const uint64_t nbytes = jb.nbytes;
    const uint32_t nfull = (uint32_t)(nbytes >> 7);
    const uint32_t rem = (uint32_t)(nbytes & 127);
    const uint64_t total = jb.total_prev + nbytes;
#if defined(SNAPHASH_EXPERIMENT_IDLE_HELPERS) // timing experiment only (wrong digests): what do the round waves cost alone?
    for (uint32_t tau = 0; tau < steps; ++tau) __syncthreads();
    (void)nfull; (void)rem; (void)total; (void)hk; (void)lane;
    return;
#endif
    uint4* __restrict__ tile = sh.tile[hk];
    const uint32_t piece = lane & 7u;
    const uint8_t* tptr[8];
    uint32_t tnp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = 8 * i + (int)(lane >> 3);
        const uint64_t dd = shfl_u64(jb.data, t);
        const uint64_t nb = shfl_u64(nbytes, t);
        tptr[i] = reinterpret_cast<const uint8_t*>(dd) + piece * 16u;
        tnp[i] = (uint32_t)((nb + 15u) >> 4);
    }
    uint4 pre[8]; // block hk, then hk+2, ... (prefetched two block-times ahead of its use)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t p = hk * 8u + piece;
        pre[i] = make_uint4(0, 0, 0, 0);
        if (p < tnp[i]) pre[i] = load_u4(tptr[i] + (uint64_t)hk * 128u);
    }
    uint64_t w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = 0;
    for (uint32_t tau = 0; tau < steps; ++tau) {
        __syncthreads();
        if ((tau & 1u) == hk) {
            // first half of block tau: stage, swap, pad, words 0..39
            const uint32_t b = tau;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) tile[(8 * i + (lane >> 3)) * kTileRow + piece] = pre[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                [[maybe_unused]] const uint32_t p = (b + 2u) * 8u + piece;
                pre[i] = make_uint4(0, 0, 0, 0);
#if !defined(SNAPHASH_EXPERIMENT_NO_FETCH) // timing experiment only (wrong digests): no global loads in the helpers
                if (p < tnp[i]) pre[i] = load_u4(tptr[i] + (uint64_t)(b + 2u) * 128u);
#endif
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint4 q = tile[lane * kTileRow + k];
                w[2 * k] = be64(q.x, q.y);
                w[2 * k + 1] = be64(q.z, q.w);
            }
            if (__any(b >= nfull && b < nblk)) apply_padding(w, b >= nfull, b - nfull, rem, total);
            schedule_span<0, 40>(w, &sh.kw[b % 3u][lane * kKwRow]);
        } else if (tau >= 1u) {
            // second half of block tau-1: words 40..79
            schedule_span<40, 80>(w, &sh.kw[(tau - 1u) % 3u][lane * kKwRow]);
        }
    }
}

template <bool PAIR>
__device__ __forceinline__ void sha512_split_body(const Job* __restrict__ jobs, uint32_t njobs, uint64_t* __restrict__ state,
                                                  uint8_t* __restrict__ digests)
{
    __shared__ SplitShared sh;
    constexpr uint32_t kRoundWaves = PAIR ? 2u : 1u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // provably wave-uniform: scalar branches
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * 64u + lane;
    const bool have = slot < njobs;

    Job jb;
    if (have) {
        jb = jobs[slot];
    } else {
        jb.data = 0; jb.nbytes = 0; jb.total_prev = 0; jb.idx = 0; jb.flags = 0;
    }
    const bool fin = (jb.flags & kJobFinal) != 0;
    const uint32_t nblk = have ? padded_blocks(jb.nbytes, fin) : 0u;

    if (threadIdx.x == 0) sh.maxblk = 0;
    __syncthreads();
    if (wave == kRoundWaves) atomicMax(&sh.maxblk, nblk); // first helper wave: lane = stream
    __syncthreads();
    const uint32_t steps = sh.maxblk + 2u; // every wave runs exactly `steps` barriers below

    if (PAIR && wave < kRoundWaves) {
        pair_round_wave(sh, wave, lane, jobs, njobs, steps, state, digests);
        return;
    }
    if (!PAIR && wave == 0) {
        // ---------------- round wave ----------------
        uint64_t H[8];
        if (jb.flags & kJobFirst) {
#pragma unroll
            for (int k = 0; k < 8; ++k) H[k] = IV512[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) H[k] = have ? state[(uint64_t)jb.idx * 8 + k] : 0;
        }
        uint32_t ring = 0; // (tau - 2) % 3
        __builtin_amdgcn_s_setprio(3); // critical path of every stream in the workgroup
        for (uint32_t tau = 0; tau < steps; ++tau) {
            __syncthreads();
            if (tau < 2) continue;
            const uint32_t b = tau - 2u;
            const uint64_t* __restrict__ row = &sh.kw[ring][lane * kKwRow];
            ring = (ring == 2u) ? 0u : ring + 1u;
            // K+W arrives 16 rounds ahead of its use: two register sets, ping-pong
            uint64_t a = H[0], bb = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
            uint64_t ka[16], kb[16];
            load_kw16(ka, row);
#pragma unroll 1
            for (int i = 0; i < 2; ++i) {
                load_kw16(kb, row + 32 * i + 16);
                rounds16(a, bb, c, d, e, f, g, h, ka);
                load_kw16(ka, row + 32 * i + 32);
                rounds16(a, bb, c, d, e, f, g, h, kb);
            }
            rounds16(a, bb, c, d, e, f, g, h, ka);
            if (b < nblk) {
                H[0] += a; H[1] += bb; H[2] += c; H[3] += d;
                H[4] += e; H[5] += f; H[6] += g; H[7] += h;
            }
        }
        if (have) {
            if (fin) {
                store_digest_be(digests + (uint64_t)jb.idx * 64, H);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) state[(uint64_t)jb.idx * 8 + k] = H[k];
            }
        }
        return;
    }

    split_helper_wave(sh, wave - kRoundWaves, lane, jb, nblk, steps);
}

template <bool PAIR>
__global__ __launch_bounds__(PAIR ? 256 : 192) void sha512_split_kernel(const Job* __restrict__ jobs, uint32_t njobs,
                                                                        uint64_t* __restrict__ state,
                                                                        uint8_t* __restrict__ digests)
{
    sha512_split_body<PAIR>(jobs, njobs, state, digests);
}

// The same lane-pair code under a second name, launched for the batches of a staged pass (files or caller memory on
// their way through the staging slots).  One command -- bench.py -- launches the kernel both ways: once over the whole
// HBM-resident tree (the roofline object, sha512_split_kernel<true>) and ~24 times per end-to-end step over a batch;
// under one name rocprofv3 --stats would average the two.
__global__ __launch_bounds__(256) void sha512_pair_staged_kernel(const Job* __restrict__ jobs, uint32_t njobs,
                                                                 uint64_t* __restrict__ state, uint8_t* __restrict__ digests)
{
    sha512_split_body<true>(jobs, njobs, state, digests);
}

#if defined(SNAPHASH_WITH_QUAD)
// ---------------------------------------------------------------------------
// QUAD kernel: as PAIR, with every stream on FOUR lanes of a round wave -- role (e-chain / a-chain) x
// half (low / high 32 bits) -- so that rotations and bitwise functions are one instruction instead of
// two; additions carry in (value, 0) register pairs and the carry crosses to the high lane once per
// round (tools/gen_quad_rounds.py -> quad_rounds.inc, proven on the CPU lane simulator,
// tests/test_quad_sim.py).  16 VALU + 1 LDS read per round instead of 19 + 0.5.  Four round waves of 16
// streams and the two helper waves per 64-stream workgroup: the helpers share SIMDs with round waves,
// which run at s_setprio 3 and are not slowed by them (profiles/r02_ubench_gfx950.txt, k_share).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void quad_round_wave(SplitShared& sh, uint32_t rw, uint32_t lane, const Job* __restrict__ jobs,
                                                uint32_t njobs, uint32_t steps, uint64_t* __restrict__ state,
                                                uint8_t* __restrict__ digests)
{
    const uint32_t j = lane & 7u;
    const bool is_hi = (lane & 4u) != 0u, is_b = (lane & 8u) != 0u;
    const uint32_t sl = 16u * rw + 4u * (lane >> 4) + (is_hi ? 7u - j : j); // stream within the workgroup
    const uint32_t slot = blockIdx.x * 64u + sl;
    const bool have = slot < njobs;
    Job jb;
    if (have) {
        jb = jobs[slot];
    } else {
        jb.data = 0; jb.nbytes = 0; jb.total_prev = 0; jb.idx = 0; jb.flags = 0;
    }
    const bool fin = (jb.flags & kJobFinal) != 0;
    const uint32_t nblk = have ? padded_blocks(jb.nbytes, fin) : 0u;
    const uint32_t word0 = is_b ? 0u : 4u; // B owns H[0..3] = a,b,c,d ; A owns H[4..7] = e,f,g,h
    // this lane's half of its four chaining words, zero-extended: the (value, 0) pair format of the block
    uint64_t h0, h1, h2, h3;
    {
        uint64_t hx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint64_t full = IV512[word0 + k];
            if (!(jb.flags & kJobFirst)) full = have ? state[(uint64_t)jb.idx * 8 + word0 + k] : 0;
            hx[k] = is_hi ? (full >> 32) : (full & 0xffffffffull);
        }
        h0 = hx[0]; h1 = hx[1]; h2 = hx[2]; h3 = hx[3];
    }
    const uint32_t c1 = is_b ? 6u : 4u, c2 = is_b ? 11u : 27u, c3 = is_b ? 28u : 14u;
    const uint32_t mb = is_b ? 0xffffffffu : 0u;
    const uint32_t zero = 0u;
    uint32_t ring = 0;
    __builtin_amdgcn_s_setprio(3);
    for (uint32_t tau = 0; tau < steps; ++tau) {
        __syncthreads();
        if (tau < 2) continue;
        const uint32_t b = tau - 2u;
        const uint64_t* row = is_b ? sh.zeros : &sh.kw[ring][sl * kKwRow];
        const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(const void*)row + (is_hi && !is_b ? 4u : 0u);
        ring = (ring == 2u) ? 0u : ring + 1u;
        asm volatile(SNAPHASH_QUAD_ROUNDS_ASM
                     : SNAPHASH_QUAD_HP0(h0), SNAPHASH_QUAD_HP1(h1), SNAPHASH_QUAD_HP2(h2), SNAPHASH_QUAD_HP3(h3)
                     : SNAPHASH_QUAD_C1(c1), SNAPHASH_QUAD_C2(c2), SNAPHASH_QUAD_C3(c3), SNAPHASH_QUAD_MB(mb),
                       SNAPHASH_QUAD_ADDR(addr), SNAPHASH_QUAD_ZERO_INPUTS(zero)
                     : SNAPHASH_QUAD_CLOBBERS, "memory");
        if (b + 1u == nblk) { // the stream ends in this block-time: store now, ride along afterwards
            const uint32_t v[4] = {(uint32_t)h0, (uint32_t)h1, (uint32_t)h2, (uint32_t)h3};
            if (fin) {
                // big-endian 64-bit words: the high half comes first
                uint32_t* o = reinterpret_cast<uint32_t*>(digests + (uint64_t)jb.idx * 64 + word0 * 8u + (is_hi ? 0u : 4u));
#pragma unroll
                for (int k = 0; k < 4; ++k) o[2 * k] = __builtin_bswap32(v[k]);
            } else {
                uint32_t* o = reinterpret_cast<uint32_t*>(state + (uint64_t)jb.idx * 8 + word0) + (is_hi ? 1 : 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[2 * k] = v[k];
            }
        }
    }
}

__global__ __launch_bounds__(384) void sha512_quad_kernel(const Job* __restrict__ jobs, uint32_t njobs,
                                                          uint64_t* __restrict__ state, uint8_t* __restrict__ digests)
{
    __shared__ SplitShared sh;
    constexpr uint32_t kRoundWaves = 4u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * 64u + lane;
    const bool have = slot < njobs;
    Job jb;
    if (have) {
        jb = jobs[slot];
    } else {
        jb.data = 0; jb.nbytes = 0; jb.total_prev = 0; jb.idx = 0; jb.flags = 0;
    }
    const uint32_t nblk = have ? padded_blocks(jb.nbytes, (jb.flags & kJobFinal) != 0) : 0u;
    if (threadIdx.x == 0) sh.maxblk = 0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kKwRow; i += 384u) sh.zeros[i] = 0;
    __syncthreads();
    if (wave == kRoundWaves) atomicMax(&sh.maxblk, nblk); // first helper wave: lane = stream
    __syncthreads();
    const uint32_t steps = sh.maxblk + 2u; // every wave runs exactly `steps` barriers below
    if (wave < kRoundWaves) {
        quad_round_wave(sh, wave, lane, jobs, njobs, steps, state, digests);
        return;
    }
    split_helper_wave(sh, wave - kRoundWaves, lane, jb, nblk, steps);
}

#endif // SNAPHASH_WITH_QUAD

// ---------------------------------------------------------------------------
// Synthetic content generator: file bytes = little-endian
// SplitMix64 stream seeded 0x5eed000000000000 ^ file_index.  SplitMix64's state
// is a plain counter, so word j of file i is mix(seed + (j+1)*gamma): fully
// parallel, one 8-byte word per thread step.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix_at(uint64_t seed, uint64_t j)
{
    uint64_t z = seed + (j + 1) * 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_synthetic_kernel(uint8_t* __restrict__ base,
                                                             const uint64_t* __restrict__ offsets,
                                                             const uint64_t* __restrict__ lens,
                                                             const uint64_t* __restrict__ findex, uint32_t nfiles)
{
    // blockIdx.y = file, blockIdx.x strides over the file's 8-byte words
    const uint32_t fi = blockIdx.y;
    if (fi >= nfiles) return;
    const uint64_t len = lens[fi];
    const uint64_t seed = 0x5eed000000000000ULL ^ findex[fi];
    uint8_t* dst = base + offsets[fi];
    const uint64_t nwords = len >> 3;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nwords;
         j += (uint64_t)gridDim.x * blockDim.x) {
        reinterpret_cast<uint64_t*>(dst)[j] = splitmix_at(seed, j);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (len & 7)) {
        const uint64_t z = splitmix_at(seed, nwords);
        for (uint32_t b = 0; b < (uint32_t)(len & 7); ++b) dst[nwords * 8 + b] = (uint8_t)(z >> (8 * b));
    }
}

// ---------------------------------------------------------------------------
// Range comparison kernel (helpers.FilesAreEqual / streamsEqual, reference
// helpers/cmp.go:31-86): pure streaming, 2 bytes read per byte compared, HBM-bound.
// One 256-thread workgroup per chunk; every lane XORs 16-byte pieces of both sides
// (coalesced dwordx4, 4 pieces in flight per lane per side), bytes past the end of
// the range are masked out of the last piece, and a differing chunk clears its
// pair's flag (all writers store 0: no ordering needed).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ranges_equal_kernel(const CmpChunk* __restrict__ chunks, uint32_t nchunks,
                                                           uint8_t* __restrict__ equal)
{
    const uint32_t ci = blockIdx.x;
    if (ci >= nchunks) return;
    const CmpChunk ch = chunks[ci];
    const uint8_t* pa = reinterpret_cast<const uint8_t*>(ch.a);
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(ch.b);
    const uint32_t npieces = (ch.nbytes + 15u) >> 4;
    const uint32_t nwhole = ch.nbytes >> 4; // pieces that lie entirely inside the range
    uint32_t diff = 0;
    uint32_t p = threadIdx.x;
    for (; p + 768u < nwhole; p += 1024u) { // 4 independent whole pieces per lane per trip
        uint4 xa[4], xb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xa[k] = load_u4(pa + (uint64_t)(p + 256u * k) * 16u);
            xb[k] = load_u4(pb + (uint64_t)(p + 256u * k) * 16u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            diff |= (xa[k].x ^ xb[k].x) | (xa[k].y ^ xb[k].y) | (xa[k].z ^ xb[k].z) | (xa[k].w ^ xb[k].w);
    }
    for (; p < npieces; p += 256u) {
        const uint4 xa = load_u4(pa + (uint64_t)p * 16u), xb = load_u4(pb + (uint64_t)p * 16u);
        uint32_t d[4] = {xa.x ^ xb.x, xa.y ^ xb.y, xa.z ^ xb.z, xa.w ^ xb.w};
        if (p == npieces - 1u && (ch.nbytes & 15u)) { // keep only the bytes that belong to the range
            const uint32_t valid = ch.nbytes & 15u;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int nv = (int)valid - 4 * w; // valid bytes in dword w
                if (nv <= 0) d[w] = 0;
                else if (nv < 4) d[w] &= (1u << (8 * nv)) - 1u;
            }
        }
        diff |= d[0] | d[1] | d[2] | d[3];
    }
    if (__any(diff != 0) && (threadIdx.x & 63u) == 0) equal[ch.pair] = 0;
}

} // namespace

hipError_t launch_compare(const CmpChunk* d_chunks, uint32_t nchunks, uint8_t* d_equal, hipStream_t s)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(ranges_equal_kernel, dim3(nchunks), dim3(256), 0, s, d_chunks, nchunks, d_equal);
    return hipGetLastError();
}

hipError_t launch_wide(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s)
{
    if (njobs == 0) return hipSuccess;
    const uint32_t grid = (njobs + 63u) / 64u;
#if defined(SNAPHASH_EXPERIMENT_WIDE_DIRECT)
    static const int form = [] { const char* e = getenv("SNAPHASH_WIDE_FORM"); return e ? atoi(e) : 0; }();
    if (form == 1) { hipLaunchKernelGGL(sha512_wide_direct_kernel<5>, dim3(grid), dim3(64), 0, s, d_jobs, njobs, d_state, d_digests); return hipGetLastError(); }
    if (form == 2) { hipLaunchKernelGGL(sha512_wide_direct_kernel<6>, dim3(grid), dim3(64), 0, s, d_jobs, njobs, d_state, d_digests); return hipGetLastError(); }
    if (form == 3) { hipLaunchKernelGGL(sha512_wide_direct_kernel<8>, dim3(grid), dim3(64), 0, s, d_jobs, njobs, d_state, d_digests); return hipGetLastError(); }
#endif
    hipLaunchKernelGGL(sha512_wide_kernel, dim3(grid), dim3(64), 0, s, d_jobs, njobs, d_state, d_digests);
    return hipGetLastError();
}

hipError_t launch_split(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s)
{
    if (njobs == 0) return hipSuccess;
    const uint32_t grid = (njobs + 63u) / 64u;
    hipLaunchKernelGGL(sha512_split_kernel<false>, dim3(grid), dim3(192), 0, s, d_jobs, njobs, d_state, d_digests);
    return hipGetLastError();
}

hipError_t launch_pair(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s, bool staged)
{
    if (njobs == 0) return hipSuccess;
    const uint32_t grid = (njobs + 63u) / 64u;
    if (staged) hipLaunchKernelGGL(sha512_pair_staged_kernel, dim3(grid), dim3(256), 0, s, d_jobs, njobs, d_state, d_digests);
    else hipLaunchKernelGGL(sha512_split_kernel<true>, dim3(grid), dim3(256), 0, s, d_jobs, njobs, d_state, d_digests);
    return hipGetLastError();
}

bool have_quad_kernel()
{
#if defined(SNAPHASH_WITH_QUAD)
    return true;
#else
    return false;
#endif
}

hipError_t launch_quad(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s)
{
#if defined(SNAPHASH_WITH_QUAD)
    if (njobs == 0) return hipSuccess;
    const uint32_t grid = (njobs + 63u) / 64u;
    hipLaunchKernelGGL(sha512_quad_kernel, dim3(grid), dim3(384), 0, s, d_jobs, njobs, d_state, d_digests);
    return hipGetLastError();
#else
    (void)d_jobs; (void)njobs; (void)d_state; (void)d_digests; (void)s;
    return hipErrorNotSupported; // snaphash_init refuses SNAPHASH_KERNEL_QUAD in a build without it
#endif
}

hipError_t launch_fill_synthetic(uint8_t* d_base, const uint64_t* d_offsets, const uint64_t* d_lens,
                                 const uint64_t* d_findex, uint32_t nfiles, uint64_t max_len, hipStream_t s)
{
    if (nfiles == 0) return hipSuccess;
    uint64_t words = (max_len >> 3) + 1;
    uint32_t gx = (uint32_t)((words + 255) / 256);
    if (gx > 64) gx = 64;
    if (gx == 0) gx = 1;
    // gridDim.y is limited to 65535: slice the file list
    for (uint32_t f0 = 0; f0 < nfiles; f0 += 65535u) {
        const uint32_t nf = (nfiles - f0 < 65535u) ? nfiles - f0 : 65535u;
        hipLaunchKernelGGL(fill_synthetic_kernel, dim3(gx, nf), dim3(256), 0, s, d_base, d_offsets + f0,
                           d_lens + f0, d_findex + f0, nf);
    }
    return hipGetLastError();
}

} // namespace snaphash
