// hostfill.cpp -- see hostfill.h.
#include "hostfill.h"

#include <ctype.h>
#include <dirent.h>
#include <errno.h>
#include <sched.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace snaphash {

namespace {

bool read_small_file(const std::string& path, std::string& out)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    out.clear();
    size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, r);
    fclose(f);
    return true;
}

// the kernel's MPOL_* numbers (linux/mempolicy.h), spelled here so that the build needs no libnuma headers
constexpr int kMpolDefault = 0, kMpolPreferred = 1;
constexpr unsigned kMpolFNode = 1u << 0, kMpolFAddr = 1u << 1;

} // namespace

std::vector<int> parse_cpulist(const std::string& text)
{
    std::vector<int> out;
    const char* p = text.c_str();
    while (*p) {
        while (*p && !isdigit((unsigned char)*p)) ++p;
        if (!*p) break;
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
        if (a < 0 || b < a || b - a > 65536) break; // not a cpulist
        for (long c = a; c <= b; ++c) out.push_back((int)c);
    }
    return out;
}

int numa_node_of_pci(const std::string& sysfs_root, const std::string& bdf_in)
{
    std::string bdf = bdf_in;
    for (char& c : bdf) c = (char)tolower((unsigned char)c);
    if (bdf.empty()) return -1;
    std::string text;
    if (!read_small_file(sysfs_root + "/bus/pci/devices/" + bdf + "/numa_node", text)) return -1;
    char* end = nullptr;
    const long v = strtol(text.c_str(), &end, 10);
    if (end == text.c_str()) return -1;
    return v < 0 ? -1 : (int)v;
}

std::vector<int> numa_cpus_of_node(const std::string& sysfs_root, int node)
{
    std::string text;
    if (node < 0 || !read_small_file(sysfs_root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", text)) return {};
    return parse_cpulist(text);
}

int numa_node_count(const std::string& sysfs_root)
{
    int n = 0;
    if (DIR* d = opendir((sysfs_root + "/devices/system/node").c_str())) {
        while (struct dirent* de = readdir(d))
            if (!strncmp(de->d_name, "node", 4) && isdigit((unsigned char)de->d_name[4])) ++n;
        closedir(d);
    }
    return n;
}

bool numa_prefer_node(int node)
{
#if defined(SYS_set_mempolicy)
    if (node < 0 || node >= 1024) return false;
    unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {0};
    mask[(size_t)node / (8 * sizeof(unsigned long))] |= 1ul << ((size_t)node % (8 * sizeof(unsigned long)));
    return syscall(SYS_set_mempolicy, kMpolPreferred, mask, (unsigned long)1024 + 1) == 0;
#else
    (void)node;
    return false;
#endif
}

void numa_default_policy()
{
#if defined(SYS_set_mempolicy)
    (void)syscall(SYS_set_mempolicy, kMpolDefault, nullptr, 0ul);
#endif
}

int numa_node_of_address(const void* addr)
{
#if defined(SYS_get_mempolicy)
    int node = -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, addr, (unsigned long)(kMpolFNode | kMpolFAddr)) != 0) return -1;
    return node;
#else
    (void)addr;
    return -1;
#endif
}

unsigned usable_cpus()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int n = CPU_COUNT(&set);
        if (n > 0) return (unsigned)n;
    }
    return std::max(1u, std::thread::hardware_concurrency());
}

// ---- FillPool ---------------------------------------------------------------------------------------------

FillPool::~FillPool() { stop(); }

void FillPool::stop()
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        quit_ = true;
    }
    cv_work_.notify_all();
    for (auto& t : th_) if (t.joinable()) t.join();
    th_.clear();
    quit_ = false;
}

void FillPool::configure(unsigned max_threads, const std::vector<int>& cpus)
{
    stop();
    max_threads_ = max_threads;
    cpus_ = cpus;
}

void FillPool::worker(unsigned id)
{
    if (!cpus_.empty()) { // the whole node's CPU set, not one CPU: the kernel balances inside it
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : cpus_) if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
        (void)sched_setaffinity(0, sizeof set, &set); // refused (cgroup cpuset without these CPUs): stay where we are
    }
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(size_t)>* fn;
        size_t n;
        std::atomic<size_t>* next;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [&] { return quit_ || (epoch_ != seen && id < want_); });
            if (quit_) return;
            seen = epoch_;
            fn = fn_; n = n_; next = next_;
        }
        for (;;) {
            const size_t i = next->fetch_add(1);
            if (i >= n) break;
            (*fn)(i);
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (--running_ == 0) cv_done_.notify_all();
        }
    }
}

void FillPool::parallel_for(size_t n, unsigned threads, const std::function<void(size_t)>& fn)
{
    if (n == 0) return;
    unsigned helpers = threads > 1 ? std::min(threads - 1, max_threads_) : 0u; // the caller works too
    if (helpers > n - 1) helpers = (unsigned)(n - 1);
    std::atomic<size_t> next{0};
    if (helpers) {
        std::lock_guard<std::mutex> lk(mu_);
        try {
            while (th_.size() < helpers) { const unsigned id = (unsigned)th_.size(); th_.emplace_back(&FillPool::worker, this, id); }
        } catch (...) { // no more threads to be had: work with the ones there are
            helpers = (unsigned)th_.size();
        }
        if (helpers) {
            fn_ = &fn; n_ = n; next_ = &next;
            want_ = helpers;
            running_ = helpers;
            ++epoch_;
        }
    }
    if (helpers) cv_work_.notify_all();
    for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n) break;
        fn(i);
    }
    if (helpers) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return running_ == 0; });
        want_ = 0;
        fn_ = nullptr; next_ = nullptr;
    }
}

// ---- copy_to_staging ------------------------------------------------------------------------------------------

#if defined(__x86_64__)
namespace {
__attribute__((target("avx2"))) void copy_nt_avx2(uint8_t* dst, const uint8_t* src, size_t n)
{
    // head: up to the first 64-byte line of the destination
    const size_t head = (64 - ((uintptr_t)dst & 63)) & 63;
    if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t lines = n / 64;
    while (lines--) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)src);
        const __m256i b = _mm256_loadu_si256((const __m256i*)(src + 32));
        _mm256_stream_si256((__m256i*)dst, a);
        _mm256_stream_si256((__m256i*)(dst + 32), b);
        src += 64; dst += 64;
    }
    n &= 63;
    if (n) memcpy(dst, src, n);
    _mm_sfence(); // the DMA engine reads this memory next: the streaming stores must be globally visible
}
const bool g_have_avx2 = __builtin_cpu_supports("avx2");
} // namespace
#endif

void copy_to_staging(void* dst, const void* src, size_t n)
{
#if defined(__x86_64__)
    if (n >= (32u << 10) && g_have_avx2) { copy_nt_avx2((uint8_t*)dst, (const uint8_t*)src, n); return; }
#endif
    memcpy(dst, src, n);
}

} // namespace snaphash
