// Aggregate rate of the library's host SHA-512 (hostsha.cpp) on T threads, each over its own buffer: what the planner's
// host side (plan_host_streams, snaphash_api.cpp) may count on per core as the thread count grows on THIS box.
//   g++ -O3 -std=c++17 -funroll-loops -pthread -o tools/hostsha_threads tools/hostsha_threads.cpp
#include "../snappy_amd/csrc/hostsha.cpp"

#include <stdio.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

using namespace snaphash;

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 32;
    for (unsigned T : {1u, 2u, 4u, 8u, 12u, 16u, 24u, 32u, 48u, 64u, 96u, 128u}) {
        std::vector<std::vector<uint8_t>> bufs(T);
        for (auto& b : bufs) { b.resize(mib << 20); for (size_t i = 0; i < b.size(); i += 4096) b[i] = (uint8_t)i; }
        std::atomic<unsigned> ready{0};
        std::atomic<bool> go{false};
        std::vector<std::thread> th;
        std::vector<double> dt(T);
        for (unsigned t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                ++ready;
                while (!go.load()) {}
                const auto t0 = std::chrono::steady_clock::now();
                HostSha hs;
                uint8_t out[64];
                host_sha512_init(hs);
                host_sha512_update(hs, bufs[t].data(), bufs[t].size());
                host_sha512_final(hs, out);
                dt[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
        while (ready.load() < T) {}
        const auto t0 = std::chrono::steady_clock::now();
        go = true;
        for (auto& x : th) x.join();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double worst = 0;
        for (double d : dt) worst = std::max(worst, d);
        printf("%3u threads x %zu MiB: wall %.1f ms, aggregate %.2f GB/s, per thread %.2f GB/s (slowest thread %.1f ms)\n", T, mib,
               wall * 1e3, T * (double)(mib << 20) / wall / 1e9, (double)(mib << 20) / worst / 1e9, worst * 1e3);
        fflush(stdout);
    }
    return 0;
}
