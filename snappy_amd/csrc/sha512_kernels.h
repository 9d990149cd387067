// sha512_kernels.h -- internal (C++) interface between the C-ABI layer and the
// HIP kernels.  Not part of the public ABI (that is include/snaphash.h).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace snaphash {

enum : uint32_t { kJobFirst = 1u, kJobFinal = 2u };

// One stream segment.  A file larger than a staging buffer is hashed as
// several segments: all but the last are multiples of 128 bytes, the chaining
// value travels between launches in state[idx].
struct Job {
    uint64_t data;       // device address of the segment's first byte (16-byte aligned)
    uint64_t nbytes;     // bytes in this segment
    uint64_t total_prev; // bytes of this stream hashed by earlier segments
    uint32_t idx;        // row in state[] / digests[]
    uint32_t flags;      // kJobFirst: start from the IV; kJobFinal: pad and emit the digest
};
static_assert(sizeof(Job) == 32, "Job is 32 bytes");

hipError_t launch_wide(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s);
// staged: the launch is one batch of a staged pass (the same code under the name sha512_pair_staged_kernel, so that a
// profile of bench.py keeps the HBM-resident launches apart from them)
hipError_t launch_pair(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s, bool staged = false);
hipError_t launch_split(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s);
hipError_t launch_quad(const Job* d_jobs, uint32_t njobs, uint64_t* d_state, uint8_t* d_digests, hipStream_t s);
bool have_quad_kernel(); // built with `make QUAD=1` only
// Byte-range comparison (helpers.FilesAreEqual, reference helpers/cmp.go:31-86, batched): one
// chunk = up to kCmpChunk bytes of one pair; equal[pair] is cleared when any chunk differs.
constexpr uint32_t kCmpChunk = 256u << 10;
struct CmpChunk {
    uint64_t a;      // device address of the chunk on the A side (16-byte aligned)
    uint64_t b;      // ... on the B side
    uint32_t nbytes; // <= kCmpChunk
    uint32_t pair;   // row of equal[]
};
static_assert(sizeof(CmpChunk) == 24, "CmpChunk is 24 bytes");
hipError_t launch_compare(const CmpChunk* d_chunks, uint32_t nchunks, uint8_t* d_equal, hipStream_t s);

hipError_t launch_fill_synthetic(uint8_t* d_base, const uint64_t* d_offsets, const uint64_t* d_lens,
                                 const uint64_t* d_findex, uint32_t nfiles, uint64_t max_len, hipStream_t s);

} // namespace snaphash
