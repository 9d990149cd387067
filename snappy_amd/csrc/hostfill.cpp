// hostfill.cpp -- see hostfill.h.
#include "hostfill.h"

#include <new>

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

std::vector<int> slice_cpus(const std::vector<int>& all, size_t pos, size_t m)
{
    if (m < 2 || pos >= m) return all;
    std::vector<int> mine;
    for (size_t a = 0; a < all.size();) {
        size_t b = a + 1;
        while (b < all.size() && all[b] == all[b - 1] + 1) ++b;
        const size_t len = b - a;
        for (size_t i = a + len * pos / m; i < a + len * (pos + 1) / m; ++i) mine.push_back(all[i]);
        a = b;
    }
    return mine;
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

bool numa_prefer_node(int node, SavedMemPolicy* saved)
{
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
    if (node < 0 || node >= 1024) return false;
    // what the thread had (numactl --membind / --interleave, the application's own set_mempolicy) comes back afterwards
    saved->valid = syscall(SYS_get_mempolicy, &saved->mode, saved->mask, (unsigned long)1024 + 1, nullptr, 0ul) == 0;
    unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {0};
    mask[(size_t)node / (8 * sizeof(unsigned long))] |= 1ul << ((size_t)node % (8 * sizeof(unsigned long)));
    return syscall(SYS_set_mempolicy, kMpolPreferred, mask, (unsigned long)1024 + 1) == 0;
#else
    (void)node; (void)saved;
    return false;
#endif
}

void numa_restore_policy(const SavedMemPolicy& saved)
{
#if defined(SYS_set_mempolicy)
    if (saved.valid && saved.mode != kMpolDefault &&
        syscall(SYS_set_mempolicy, saved.mode, saved.mask, (unsigned long)1024 + 1) == 0)
        return;
    (void)syscall(SYS_set_mempolicy, kMpolDefault, nullptr, 0ul);
#else
    (void)saved;
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

// CPU quota of the cgroup(s) this process sits in, in whole CPUs (0 = none found).  A 1-GPU slice of a big host often
// sees every CPU in its affinity mask and a CFS quota of 16: threads beyond the quota are throttled, not run.
// cgroup v2: cpu.max = "<quota|max> <period>" in the process's group and every ancestor; v1: cpu.cfs_quota_us / _period_us.
unsigned cgroup_cpu_quota(const std::string& sysfs_cgroup_root, const std::string& proc_self_cgroup)
{
    std::string text;
    if (!read_small_file(proc_self_cgroup, text)) return 0;
    double best = 0;
    auto take = [&](double quota, double period) {
        if (quota > 0 && period > 0) { const double c = quota / period; if (best == 0 || c < best) best = c; }
    };
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        const std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        const size_t c1 = line.find(':'), c2 = c1 == std::string::npos ? c1 : line.find(':', c1 + 1);
        if (c2 == std::string::npos) continue;
        const std::string ctrl = line.substr(c1 + 1, c2 - c1 - 1);
        std::string path = line.substr(c2 + 1);
        const bool v2 = ctrl.empty();
        if (!v2) { // v1: the hierarchy that carries the "cpu" controller ("cpu", "cpu,cpuacct")
            bool has_cpu = false;
            for (size_t a = 0; a <= ctrl.size();) {
                size_t b = ctrl.find(',', a);
                if (b == std::string::npos) b = ctrl.size();
                if (ctrl.compare(a, b - a, "cpu") == 0) has_cpu = true;
                a = b + 1;
            }
            if (!has_cpu) continue;
        }
        const std::string base = sysfs_cgroup_root + (v2 ? "" : "/" + ctrl);
        for (;;) { // the group and its ancestors: the tightest quota binds
            std::string v;
            if (v2) {
                if (read_small_file(base + path + "/cpu.max", v) && v.compare(0, 3, "max") != 0) {
                    char* end = nullptr;
                    const double q = strtod(v.c_str(), &end);
                    const double per = end ? strtod(end, nullptr) : 0;
                    take(q, per);
                }
            } else {
                std::string pv;
                if (read_small_file(base + path + "/cpu.cfs_quota_us", v) && read_small_file(base + path + "/cpu.cfs_period_us", pv))
                    take(strtod(v.c_str(), nullptr), strtod(pv.c_str(), nullptr));
            }
            if (path.empty() || path == "/") break;
            const size_t slash = path.rfind('/');
            path = slash == 0 || slash == std::string::npos ? "/" : path.substr(0, slash);
        }
    }
    if (best <= 0) return 0;
    return (unsigned)std::max(1.0, best + 0.5);
}

unsigned usable_cpus()
{
    unsigned n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0) n = (unsigned)c;
    }
    if (n == 0) n = std::max(1u, std::thread::hardware_concurrency());
    static const unsigned quota = cgroup_cpu_quota("/sys/fs/cgroup", "/proc/self/cgroup");
    if (quota && quota < n) n = quota;
    return n;
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
        // ... of those the process itself may run on: a taskset / sched_setaffinity restriction the application was
        // started with is never widened
        cpu_set_t set, allowed;
        CPU_ZERO(&set);
        CPU_ZERO(&allowed);
        const bool have_allowed = sched_getaffinity(0, sizeof allowed, &allowed) == 0;
        for (int c : cpus_)
            if (c >= 0 && c < CPU_SETSIZE && (!have_allowed || CPU_ISSET(c, &allowed))) CPU_SET(c, &set);
        if (CPU_COUNT(&set) > 0) (void)sched_setaffinity(0, sizeof set, &set); // refused (cgroup cpuset without these CPUs): stay where we are
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
        try {
            for (;;) {
                const size_t i = next->fetch_add(1);
                if (i >= n) break;
                (*fn)(i);
            }
        } catch (...) { // (not on the caller's stack: remembered, raised there)
            fn_threw_.store(true);
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
    bool threw = false;
    try {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) break;
            fn(i);
        }
    } catch (...) { // the helpers still hold fn and next, which live on this stack: stop them, wait, then raise
        threw = true;
        next.store(n);
    }
    if (helpers) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return running_ == 0; });
        want_ = 0;
        fn_ = nullptr; next_ = nullptr;
    }
    if (fn_threw_.exchange(false) || threw) throw std::bad_alloc();
}

// ---- run_on_threads: the process-wide helper pool ----------------------------------------------------------------

void run_on_threads(unsigned T, const std::function<void(unsigned)>& fn)
{
    if (T <= 1) { fn(0); return; }
    static std::mutex mu;
    static FillPool* pool = [] { // (never destroyed: its threads sleep until the process ends; a static destructor would race with them at exit)
        FillPool* p = new FillPool();
        p->configure(64, {});
        return p;
    }();
    static const pid_t born_in = getpid(); // a forked child inherits the pool's object but none of its threads: it starts its own, per call
    if (T <= 65 && getpid() == born_in && mu.try_lock()) {
        std::lock_guard<std::mutex> lk(mu, std::adopt_lock);
        // parallel_for hands out indices, not threads: an index is taken by whoever comes first, so give every index a
        // thread's whole share of the work -- which is what fn(t) is
        pool->parallel_for(T, T, [&fn](size_t t) { fn((unsigned)t); });
        return;
    }
    ThreadJoiner th;
    for (unsigned t = 1; t < T; ++t) th.spawn(fn, t);
    fn(0);
    th.join_all();
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
