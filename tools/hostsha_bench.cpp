// Rate of the library's host SHA-512 block function, every spelling the CPU supports (hostsha.cpp), best of 25 passes
// over 8 MiB.  Build on the machine to be measured:
//   g++ -O3 -std=c++17 -funroll-loops [-DHS_ASSOC=0] -o /tmp/hostsha_bench tools/hostsha_bench.cpp
#include "../snappy_amd/csrc/hostsha.cpp"

#include <stdio.h>

#include <chrono>
#include <vector>

using namespace snaphash;

int main()
{
    std::vector<uint8_t> buf(8 << 20);
    for (size_t i = 0; i < buf.size(); ++i) buf[i] = (uint8_t)(i * 2654435761u >> 13);
    static const char* names[] = {"portable/bmi2", "avx2 schedule", "avx512vl schedule", "avx512vl schedule + a-chain"};
    for (int v = 0; v < host_sha512_variants(); ++v) {
        uint64_t H[8];
        double best = 1e9;
        for (int rep = 0; rep < 25; ++rep) {
            for (int k = 0; k < 8; ++k) H[k] = IV512[k];
            const auto t0 = std::chrono::steady_clock::now();
            host_sha512_blocks_variant(v, H, buf.data(), buf.size() / 128);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        printf("HS_ASSOC=%d %-28s %.3f GB/s  (H0 %016llx)\n", HS_ASSOC, names[v], buf.size() / best / 1e9, (unsigned long long)H[0]);
    }
    return 0;
}
