// openat_probe.cpp -- what does a file's open cost when twelve threads are at it, by full path and relative to its
// directory's descriptor?  100 000 files of 8 KiB in 1 000 directories on tmpfs (five path components), each opened,
// read and closed once, in shuffled order (as a batch sorted by size meets them).
// build: g++ -O2 -std=c++17 tools/openat_probe.cpp -o tools/openat_probe -pthread     usage: openat_probe [DIR=/dev/shm/oaprobe/a/b]
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <random>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const std::string root = argc > 1 ? argv[1] : "/dev/shm/oaprobe/a/b";
    std::string mk;
    for (size_t i = 1; i <= root.size(); ++i)
        if (i == root.size() || root[i] == '/') { mk = root.substr(0, i); mkdir(mk.c_str(), 0755); }
    const int nd = 1000, nf = 100;
    std::vector<char> buf(8192, 'x');
    std::vector<std::string> dirs(nd), paths, bases;
    std::vector<int> dir_of;
    for (int d = 0; d < nd; ++d) {
        char t[32];
        snprintf(t, sizeof t, "/d%04d", d);
        dirs[d] = root + t;
        mkdir(dirs[d].c_str(), 0755);
        for (int f = 0; f < nf; ++f) {
            snprintf(t, sizeof t, "f%06d.bin", d * nf + f);
            bases.push_back(t);
            paths.push_back(dirs[d] + "/" + t);
            dir_of.push_back(d);
            int fd = open(paths.back().c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
            if (fd < 0 || write(fd, buf.data(), buf.size()) != (ssize_t)buf.size()) { perror("write"); return 1; }
            close(fd);
        }
    }
    std::vector<uint32_t> order(paths.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
    std::shuffle(order.begin(), order.end(), std::mt19937(5));
    std::vector<int> dfd(nd);
    for (int d = 0; d < nd; ++d) dfd[d] = open(dirs[d].c_str(), O_PATH | O_DIRECTORY | O_CLOEXEC);
    printf("# %zu files of 8 KiB under %s, each opened + read + closed once, shuffled\n", paths.size(), root.c_str());
    for (unsigned T : {1u, 4u, 12u, 16u})
        for (int mode = 0; mode < 3; ++mode) { // 0: open(path)  1: openat(dirfd, base)  2: lstat(path) vs 3: fstatat
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                std::atomic<size_t> next{0};
                const double t0 = now();
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&] {
                        std::vector<char> b(8192);
                        struct stat st;
                        for (size_t k; (k = next.fetch_add(64)) < order.size();)
                            for (size_t q = k; q < std::min(k + 64, order.size()); ++q) {
                                const uint32_t i = order[q];
                                if (mode == 2) { lstat(paths[i].c_str(), &st); fstatat(dfd[dir_of[i]], bases[i].c_str(), &st, AT_SYMLINK_NOFOLLOW); continue; }
                                const int fd = mode == 0 ? open(paths[i].c_str(), O_RDONLY | O_CLOEXEC) : openat(dfd[dir_of[i]], bases[i].c_str(), O_RDONLY | O_CLOEXEC);
                                if (fd < 0) { perror("open"); continue; }
                                if (pread(fd, b.data(), b.size(), 0) < 0) perror("pread");
                                close(fd);
                            }
                    });
                for (auto& x : th) x.join();
                best = std::min(best, now() - t0);
            }
            if (mode == 2) continue;
            printf("%2u threads, %-24s %.1f ms = %.2f us per file per thread\n", T, mode == 0 ? "open(full path)" : "openat(dirfd, base)", best * 1e3,
                   best * 1e6 * T / paths.size());
        }
    // lstat against fstatat, separately
    for (unsigned T : {1u, 16u})
        for (int mode = 0; mode < 2; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                std::atomic<size_t> next{0};
                const double t0 = now();
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&] {
                        struct stat st;
                        for (size_t k; (k = next.fetch_add(64)) < order.size();)
                            for (size_t q = k; q < std::min(k + 64, order.size()); ++q) {
                                const uint32_t i = q; // in directory order, as the walk meets them
                                if (mode == 0) lstat(paths[i].c_str(), &st);
                                else fstatat(dfd[dir_of[i]], bases[i].c_str(), &st, AT_SYMLINK_NOFOLLOW);
                            }
                    });
                for (auto& x : th) x.join();
                best = std::min(best, now() - t0);
            }
            printf("%2u threads, %-24s %.1f ms = %.2f us per entry per thread\n", T, mode == 0 ? "lstat(full path)" : "fstatat(dirfd, base)", best * 1e3,
                   best * 1e6 * T / paths.size());
        }
    for (auto& p : paths) unlink(p.c_str());
    for (auto& d : dirs) rmdir(d.c_str());
    return 0;
}
