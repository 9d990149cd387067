// ubench_halfexec.hip -- does a wave with only its lower lanes active issue VALU instructions faster on gfx950?
// (If a 64-wide operation whose upper 16-lane groups are all inactive skipped their passes, a SHA-512 round wave
// that keeps 16 streams instead of 32 would advance each of them faster: DESIGN.md sec. 4.)  One wave per CU,
// a dependent chain of 16 x 8192 instructions, lanes >= `active` leave before the loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_halfexec tools/ubench_halfexec.hip && tools/ubench_halfexec
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kIters = 8192;
#define R16(I) I I I I I I I I I I I I I I I I

template <int KIND>
__global__ __launch_bounds__(64) void chain(uint32_t active, uint32_t* sink)
{
    if (threadIdx.x >= active) return;
    uint32_t r = threadIdx.x * 3 + 1, s = threadIdx.x ^ 0x5a5a;
    uint64_t q = r, p = s;
    for (int it = 0; it < kIters; ++it) {
        if (KIND == 0) asm volatile(R16("v_alignbit_b32 %0, %0, %1, 7\n") : "+v"(r) : "v"(s));
        if (KIND == 1) asm volatile(R16("v_lshl_add_u64 %0, %0, 0, %1\n") : "+v"(q) : "v"(p));
        if (KIND == 2) asm volatile(R16("v_xor_b32 %0, %0, %1\n") : "+v"(r) : "v"(s));
    }
    if ((r ^ (uint32_t)q) == 0x12345678) sink[0] = r;
}

template <int KIND>
static void run(const char* name, uint32_t* d_sink)
{
    for (uint32_t active : {64u, 48u, 32u, 16u, 1u}) {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(chain<KIND>, dim3(256), dim3(64), 0, 0, active, d_sink);
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(chain<KIND>, dim3(256), dim3(64), 0, 0, active, d_sink);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-16s active lanes %2u: %.3f ms = %.2f ns per instruction\n", name, active, ms, ms * 1e6 / (16.0 * kIters));
    }
}

int main()
{
    uint32_t* d_sink;
    CHECK(hipMalloc(&d_sink, 64));
    run<0>("v_alignbit_b32", d_sink);
    run<1>("v_lshl_add_u64", d_sink);
    run<2>("v_xor_b32", d_sink);
    return 0;
}
