// sha512_kernels.hip -- multi-buffer SHA-512 for gfx950 (MI355X, CDNA4).
//
// The GPU side of helpers.Sha512sum (reference helpers/helpers.go:187-201)
// batched over the files of one writeHashes pass (snappy/build.go:228-259).
// A file's blocks are strictly sequential (Merkle-Damgard chaining), so the
// parallel axis is the file list: every kernel here advances many independent
// streams in lockstep.
//
// Data layout in HBM: file bytes are contiguous, 16-byte aligned, described by
// one 32-byte Job per stream segment.  A wave fetches 128-byte blocks with
// full-line coalesced dwordx4 loads (8 lanes per stream, 8 streams per load
// instruction), stages them in a wave-private LDS tile (144-byte row stride to
// spread ds_read_b128 over the banks) and each lane then reads its own block.
// One block is prefetched into registers while the previous one is hashed.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// chip (>= ~130k lanes): every VALU instruction advances 64 streams.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sha512_wide_kernel(const Job* __restrict__ jobs, uint32_t njobs,
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

} // namespace

hipError_t launch_wide(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s)
{
    if (njobs == 0) return hipSuccess;
    const uint32_t grid = (njobs + 63u) / 64u;
    hipLaunchKernelGGL(sha512_wide_kernel, dim3(grid), dim3(64), 0, s, d_jobs, njobs, d_state, d_digests);
    return hipGetLastError();
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
