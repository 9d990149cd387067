// hostregister_probe.cpp -- can the DMA engines read a tree's files straight out of the page cache?
//
// The staged pass copies every byte once on the CPU (pread into pinned memory: 12 threads keep one PCIe link busy;
// DESIGN.md sec. 3) -- 3-4 bytes of DRAM traffic per byte hashed and most of the box's CPU quota.  The alternative this
// probe prices: mmap each file, hipHostRegister the mapping, hipMemcpyAsync from it.  Per file of 1 MiB that is one
// mmap, one register (pin 256 pages, map them for the GPU), one unregister, one munmap.
//
// build: hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/hostregister_probe.cpp -o tools/hostregister_probe -pthread
// usage: hostregister_probe [DIR=/dev/shm/hrprobe] [FILES=2048] [KIB=1024]
#include <fcntl.h>
#include <hip/hip_runtime.h>
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

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void sum_kernel(const uint32_t* p, size_t nwords, unsigned long long* out)
{
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    atomicAdd(out, s);
}

struct Variant {
    const char* name;
    int open_flags, prot, map_flags;
    unsigned reg_flags;
};

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm/hrprobe";
    const size_t nfiles = argc > 2 ? (size_t)atol(argv[2]) : 2048;
    const size_t len = (argc > 3 ? (size_t)atol(argv[3]) : 1024) << 10;
    mkdir(dir.c_str(), 0755);
    std::vector<std::string> paths(nfiles);
    unsigned long long want = 0;
    {
        std::vector<uint32_t> buf(len / 4);
        for (size_t f = 0; f < nfiles; ++f) {
            paths[f] = dir + "/f" + std::to_string(f);
            for (size_t i = 0; i < buf.size(); ++i) buf[i] = (uint32_t)(i * 2654435761u + f * 40503u);
            for (uint32_t w : buf) want += w;
            int fd = open(paths[f].c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
            if (fd < 0 || write(fd, buf.data(), len) != (ssize_t)len) { perror("write"); return 1; }
            close(fd);
        }
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# %zu files x %zu KiB in %s; device %s: hostRegisterSupported %d, hostRegisterReadOnlySupported %d, pageableMemoryAccess %d\n", nfiles,
           len >> 10, dir.c_str(), prop.name, prop.hostRegisterSupported, prop.hostRegisterReadOnlySupported, prop.pageableMemoryAccess);
    uint8_t* dev = nullptr;
    unsigned long long* dsum = nullptr;
    if (hipMalloc(&dev, nfiles * len) != hipSuccess || hipMalloc(&dsum, 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipStream_t st;
    hipStreamCreate(&st);

    // the reference point: pread into one pinned buffer, one big copy (what the engine does, without its pipelining)
    {
        uint8_t* pin = nullptr;
        if (hipHostMalloc(&pin, nfiles * len, hipHostMallocDefault) == hipSuccess) {
            for (unsigned T : {1u, 8u, 12u}) {
                const double t0 = now();
                std::atomic<size_t> next{0};
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&] {
                        for (size_t f; (f = next.fetch_add(1)) < nfiles;) {
                            int fd = open(paths[f].c_str(), O_RDONLY);
                            if (pread(fd, pin + f * len, len, 0) != (ssize_t)len) perror("pread");
                            close(fd);
                        }
                    });
                for (auto& x : th) x.join();
                const double t1 = now();
                hipMemcpyAsync(dev, pin, nfiles * len, hipMemcpyHostToDevice, st);
                hipStreamSynchronize(st);
                const double t2 = now();
                printf("pread into pinned, %2u threads: fill %.1f ms (%.1f GB/s), copy %.1f ms (%.1f GB/s)\n", T, (t1 - t0) * 1e3, nfiles * len / (t1 - t0) / 1e9,
                       (t2 - t1) * 1e3, nfiles * len / (t2 - t1) / 1e9);
            }
            hipHostFree(pin);
        }
    }

    const Variant variants[] = {
        {"O_RDONLY, PROT_READ, MAP_SHARED, register Default", O_RDONLY, PROT_READ, MAP_SHARED, hipHostRegisterDefault},
        {"O_RDONLY, PROT_READ, MAP_SHARED, register ReadOnly", O_RDONLY, PROT_READ, MAP_SHARED, hipHostRegisterReadOnly},
        {"O_RDONLY, PROT_READ, MAP_SHARED|MAP_POPULATE, register ReadOnly", O_RDONLY, PROT_READ, MAP_SHARED | MAP_POPULATE, hipHostRegisterReadOnly},
        {"O_RDWR, PROT_READ|PROT_WRITE, MAP_SHARED, register Default", O_RDWR, PROT_READ | PROT_WRITE, MAP_SHARED, hipHostRegisterDefault},
        {"O_RDWR, PROT_READ|PROT_WRITE, MAP_SHARED|MAP_POPULATE, register Default", O_RDWR, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, hipHostRegisterDefault},
    };
    for (const Variant& v : variants) {
        printf("## %s\n", v.name);
        fflush(stdout);
        bool works = true;
        for (unsigned T : {1u, 4u, 8u, 16u}) {
            if (!works) break;
            std::vector<void*> maps(nfiles, nullptr);
            std::atomic<size_t> next{0};
            std::atomic<int> failed{0};
            std::atomic<long long> ns_map{0}, ns_reg{0};
            const double t0 = now();
            {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&] {
                        hipSetDevice(0);
                        for (size_t f; (f = next.fetch_add(1)) < nfiles;) {
                            if (failed.load()) break;
                            const double a = now();
                            int fd = open(paths[f].c_str(), v.open_flags);
                            void* p = fd >= 0 ? mmap(nullptr, len, v.prot, v.map_flags, fd, 0) : MAP_FAILED;
                            if (fd >= 0) close(fd);
                            if (p == MAP_FAILED) { failed = 1; perror("mmap"); break; }
                            const double b = now();
                            hipError_t e = hipHostRegister(p, len, v.reg_flags);
                            const double c = now();
                            if (e != hipSuccess) {
                                if (!failed.exchange(2)) printf("hipHostRegister: %s\n", hipGetErrorString(e));
                                munmap(p, len);
                                (void)hipGetLastError();
                                break;
                            }
                            maps[f] = p;
                            ns_map += (long long)((b - a) * 1e9);
                            ns_reg += (long long)((c - b) * 1e9);
                        }
                    });
                for (auto& x : th) x.join();
            }
            const double t1 = now();
            if (failed.load()) works = false;
            double t2 = t1, t3 = t1;
            bool copied = false;
            if (works) {
                hipError_t e = hipSuccess;
                for (size_t f = 0; f < nfiles && e == hipSuccess; ++f) e = hipMemcpyAsync(dev + f * len, maps[f], len, hipMemcpyHostToDevice, st);
                t2 = now();
                hipError_t e2 = hipStreamSynchronize(st);
                t3 = now();
                if (e != hipSuccess || e2 != hipSuccess) {
                    printf("copy from the registered mappings: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
                    works = false;
                } else {
                    copied = true;
                }
            }
            unsigned long long got = 0;
            if (copied) {
                hipMemsetAsync(dsum, 0, 8, st);
                hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, st, (const uint32_t*)dev, nfiles * len / 4, dsum);
                hipMemcpyAsync(&got, dsum, 8, hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
                hipMemsetAsync(dev, 0, nfiles * len, st);
                hipStreamSynchronize(st);
            }
            const double t4 = now();
            next = 0;
            {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&] {
                        hipSetDevice(0);
                        for (size_t f; (f = next.fetch_add(1)) < nfiles;) {
                            if (!maps[f]) continue;
                            hipHostUnregister(maps[f]);
                            munmap(maps[f], len);
                        }
                    });
                for (auto& x : th) x.join();
            }
            const double t5 = now();
            if (copied)
                printf("%2u threads: mmap+register %.1f ms wall (%.1f us mmap, %.1f us register per file per thread) = %.1f GB/s of files; "
                       "enqueue %.1f ms, copies done after %.1f ms (%.1f GB/s); sum %s; unregister+munmap %.1f ms wall\n",
                       T, (t1 - t0) * 1e3, ns_map / 1e3 / nfiles, ns_reg / 1e3 / nfiles, nfiles * len / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t1) * 1e3,
                       nfiles * len / (t3 - t1) / 1e9, got == want ? "ok" : "WRONG", (t5 - t4) * 1e3);
            fflush(stdout);
        }
    }
    for (auto& p : paths) unlink(p.c_str());
    rmdir(dir.c_str());
    hipFree(dev);
    hipFree(dsum);
    return 0;
}
