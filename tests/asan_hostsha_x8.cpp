// Test-only driver: snappy_amd/csrc/hostsha_x8.cpp (eight streams side by side on one core) under AddressSanitizer + UBSan
// (CPU build; GPU sanitizers are not available on the pool).  Random sets of memory streams and of files -- lengths around
// the block, the 256 KiB chunk and beyond, 1..8 lanes, lanes that empty and refill at different times -- against the
// one-stream code of hostsha.cpp.  Exit code 0 = no sanitizer report, every digest equal.   usage: asan_hostsha_x8 TMPDIR
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <sys/resource.h>

#include <atomic>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../snappy_amd/csrc/hostsha.h"

using namespace snaphash;

static void one(const uint8_t* p, size_t n, uint8_t* out)
{
    HostSha s;
    host_sha512_init(s);
    host_sha512_update(s, p, n);
    host_sha512_final(s, out);
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    std::mt19937_64 rng(5);
    auto pick_len = [&]() -> size_t {
        switch (rng() % 8) {
        case 0: return rng() % 300;
        case 1: return 128 * (rng() % 50);
        case 2: return (256u << 10) * (1 + rng() % 3) + (rng() % 3) * 128 - (rng() % 2);
        case 3: return 0;
        default: return rng() % 700000;
        }
    };
    for (int trial = 0; trial < 120; ++trial) {
        const int n = 1 + (int)(rng() % 20);
        const bool files = trial % 3 == 2;
        std::vector<std::vector<uint8_t>> bufs(n);
        std::vector<std::string> paths(n);
        std::vector<uint8_t> dig(64 * n), ref(64 * n);
        for (int i = 0; i < n; ++i) {
            bufs[i].resize(pick_len());
            for (auto& b : bufs[i]) b = (uint8_t)rng();
            one(bufs[i].data(), bufs[i].size(), ref.data() + 64 * i);
            if (files) {
                paths[i] = std::string(argv[1]) + "/f" + std::to_string(trial) + "_" + std::to_string(i);
                const int fd = open(paths[i].c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
                if (fd < 0 || (bufs[i].size() && write(fd, bufs[i].data(), bufs[i].size()) != (ssize_t)bufs[i].size())) return 3;
                close(fd);
            }
        }
        int64_t nexti = 0, bad = -1;
        const unsigned lanes = 1 + (unsigned)(rng() % 8);
        const uint64_t alone_from = trial % 4 == 0 ? 400000 : 0;
        const int rc = host_sha512_many(
            lanes, [&]() -> int64_t { return nexti < n ? nexti++ : -1; },
            [&](int64_t id) {
                HostStream h;
                if (files) h.path = paths[id].c_str();
                else h.mem = bufs[id].data();
                h.len = bufs[id].size();
                h.digest = dig.data() + 64 * id;
                h.alone = alone_from && h.len >= alone_from;
                return h;
            },
            &bad);
        if (rc || memcmp(dig.data(), ref.data(), 64 * (size_t)n) != 0) {
            printf("trial %d: rc %d, lanes %u\n", trial, rc, lanes);
            return 4;
        }
        if (files)
            for (auto& p : paths) unlink(p.c_str());
    }
    // Out of descriptors: six threads x eight lanes want 48 descriptors where the soft limit leaves about ten.
    // An open that meets EMFILE puts its stream aside (one at a time when the lanes have drained, waiting for a
    // descriptor): every digest must still come out, as the reference's one-file-at-a-time loop would have it.
    {
        const int n = 160, T = 6;
        std::vector<std::vector<uint8_t>> bufs(n);
        std::vector<std::string> paths(n);
        std::vector<uint8_t> dig(64 * n), ref(64 * n);
        for (int i = 0; i < n; ++i) {
            bufs[i].resize(1000 + rng() % 400000);
            for (auto& b : bufs[i]) b = (uint8_t)rng();
            one(bufs[i].data(), bufs[i].size(), ref.data() + 64 * i);
            paths[i] = std::string(argv[1]) + "/e_" + std::to_string(i);
            const int fd = open(paths[i].c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
            if (fd < 0 || write(fd, bufs[i].data(), bufs[i].size()) != (ssize_t)bufs[i].size()) return 3;
            close(fd);
        }
        int held = 0;
        for (int fd = 0; fd < 256; ++fd) held += fcntl(fd, F_GETFD) != -1;
        struct rlimit rl, low;
        if (getrlimit(RLIMIT_NOFILE, &rl) != 0) return 5;
        low = rl;
        low.rlim_cur = (rlim_t)(held + 10);
        std::atomic<int64_t> nexti{0};
        std::atomic<int> bad_rc{0}, go{0}, finished{0}, leave{0};
        std::vector<std::thread> th;
        // (the threads start before the limit drops and end after it is back: the sanitizer's own checks at thread entry and
        // exit read /proc, and report nonsense when they find no descriptor for that)
        for (int t = 0; t < T; ++t)
            th.emplace_back([&] {
                while (!go.load()) std::this_thread::yield();
                int64_t bad = -1;
                const int rc = host_sha512_many(
                    8, [&]() -> int64_t { const int64_t k = nexti.fetch_add(1); return k < n ? k : -1; },
                    [&](int64_t id) {
                        HostStream h;
                        h.path = paths[id].c_str();
                        h.len = bufs[id].size();
                        h.digest = dig.data() + 64 * id;
                        return h;
                    },
                    &bad);
                if (rc) bad_rc.store(rc);
                finished.fetch_add(1);
                while (!leave.load()) std::this_thread::yield();
            });
        if (setrlimit(RLIMIT_NOFILE, &low) != 0) return 5;
        go.store(1);
        while (finished.load() < T) std::this_thread::yield();
        setrlimit(RLIMIT_NOFILE, &rl);
        leave.store(1);
        for (auto& t : th) t.join();
        for (auto& p : paths) unlink(p.c_str());
        if (bad_rc.load() || memcmp(dig.data(), ref.data(), 64 * (size_t)n) != 0) {
            printf("descriptor-starved pass: rc %d (%s)\n", bad_rc.load(), strerror(bad_rc.load()));
            return 6;
        }
    }
    printf("asan x8 driver ok (x8 %s)\n", host_sha512_x8_available() ? "used" : "not available: the one-stream code ran");
    return 0;
}
