// mmap_fill_probe.cpp -- is pread the cheapest way to bring a file's bytes from the page cache into a staging buffer?
// The staging fill costs 12 threads to keep one PCIe link busy (6.5 GB/s a thread); every core it frees
// would hash (hostsha_x8.cpp).  Compared, per thread count, over 2 048 files of 1 MiB on tmpfs into one big buffer:
//   pread                     (what the engine does)
//   mmap + memcpy + munmap    (MAP_SHARED, faults taken as they come)
//   mmap(MAP_POPULATE) + non-temporal copy + munmap
// build: g++ -O2 -std=c++17 -mavx2 tools/mmap_fill_probe.cpp -o tools/mmap_fill_probe -pthread     usage: mmap_fill_probe [DIR]
#include <fcntl.h>
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_nt(uint8_t* dst, const uint8_t* src, size_t n)
{
    for (size_t i = 0; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i*)(dst + i), _mm256_loadu_si256((const __m256i*)(src + i)));
    _mm_sfence();
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm/mmprobe";
    const size_t nfiles = 2048, len = 1u << 20;
    mkdir(dir.c_str(), 0755);
    std::vector<std::string> paths(nfiles);
    {
        std::vector<uint8_t> b(len);
        for (size_t f = 0; f < nfiles; ++f) {
            paths[f] = dir + "/f" + std::to_string(f);
            for (size_t i = 0; i < len; i += 64) b[i] = (uint8_t)(i + f);
            int fd = open(paths[f].c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
            if (fd < 0 || write(fd, b.data(), len) != (ssize_t)len) { perror("write"); return 1; }
            close(fd);
        }
    }
    uint8_t* dst = (uint8_t*)aligned_alloc(4096, nfiles * len);
    memset(dst, 1, nfiles * len);
    printf("# %zu files x 1 MiB under %s -> one 2 GiB buffer; best of 3\n", nfiles, dir.c_str());
    const char* names[] = {"pread", "mmap + memcpy + munmap", "mmap(MAP_POPULATE) + non-temporal copy + munmap"};
    for (unsigned T : {1u, 6u, 12u, 16u})
        for (int mode = 0; mode < 3; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                std::atomic<size_t> next{0};
                const double t0 = now();
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&] {
                        for (size_t f; (f = next.fetch_add(1)) < nfiles;) {
                            const int fd = open(paths[f].c_str(), O_RDONLY | O_CLOEXEC);
                            if (fd < 0) { perror("open"); continue; }
                            if (mode == 0) {
                                if (pread(fd, dst + f * len, len, 0) != (ssize_t)len) perror("pread");
                            } else {
                                void* p = mmap(nullptr, len, PROT_READ, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0);
                                if (p == MAP_FAILED) { perror("mmap"); close(fd); continue; }
                                if (mode == 1) memcpy(dst + f * len, p, len);
                                else copy_nt(dst + f * len, (const uint8_t*)p, len);
                                munmap(p, len);
                            }
                            close(fd);
                        }
                    });
                for (auto& x : th) x.join();
                best = std::min(best, now() - t0);
            }
            printf("%2u threads, %-48s %.1f ms = %.1f GB/s (%.1f GB/s a thread)\n", T, names[mode], best * 1e3, nfiles * len / best / 1e9, nfiles * len / best / 1e9 / T);
        }
    for (auto& p : paths) unlink(p.c_str());
    rmdir(dir.c_str());
    free(dst);
    return 0;
}
