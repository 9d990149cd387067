// hostfill.h -- the host side of an engine's staging: where its pinned memory lives and which threads fill it.
// Internal.  No reference counterpart: helpers.Sha512sum (helpers/helpers.go:187-201) reads through a 32 KiB
// io.Copy buffer on the calling goroutine; here the bytes of a whole batch are moved into pinned memory by a pool
// of threads that sits on the GPU's own NUMA node, because with one engine per GPU eight engines
// move 8 x 55 GB/s and a staging buffer on the far socket would put every byte on the inter-socket link twice.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace snaphash {

// ---- NUMA topology from sysfs (root = "/sys"; the tests hand in a fake tree) ----------------------------------
// NUMA node of the PCI function `bdf` ("0000:c1:00.0", as hipDeviceGetPCIBusId spells it), -1 when sysfs does not say
// (single-node boxes report -1).
int numa_node_of_pci(const std::string& sysfs_root, const std::string& bdf);
// CPUs of a node (<root>/devices/system/node/node<N>/cpulist, "0-15,128-143"); empty when unknown.
std::vector<int> numa_cpus_of_node(const std::string& sysfs_root, int node);
std::vector<int> parse_cpulist(const std::string& text);
int numa_node_count(const std::string& sysfs_root);
// Engines that share a NUMA node share its CPUs: engine `pos` of the `m` on a node gets slice `pos` of every contiguous
// run of the node's cpulist (a run is one SMT sibling set on the hosts seen: "0-63,128-191"), so the fill threads of two
// engines never sit on one core and a core's two hardware threads stay with one engine.
std::vector<int> slice_cpus(const std::vector<int>& all, size_t pos, size_t m);

// Memory policy of the calling thread: prefer `node` for the allocations that follow (set_mempolicy MPOL_PREFERRED;
// no libnuma: the raw system calls), then back to what the thread had before -- the caller's own policy (numactl,
// set_mempolicy) is read first and restored, not reset to the default.  false when the kernel refuses
// (no NUMA, seccomp).
struct SavedMemPolicy {
    bool valid = false;
    int mode = 0;
    unsigned long mask[1024 / (8 * sizeof(unsigned long)) + 1] = {0};
};
bool numa_prefer_node(int node, SavedMemPolicy* saved);
void numa_restore_policy(const SavedMemPolicy& saved);
// Node the page holding `addr` lives on (get_mempolicy MPOL_F_NODE | MPOL_F_ADDR), -1 when unknown.
int numa_node_of_address(const void* addr);
// CPUs the process may keep busy: its affinity mask, capped by the CFS quota of its cgroup (the GPU box gives a 1-GPU
// slice all 256 CPUs in the mask and a quota of 16: the 17th busy thread is throttled).
unsigned usable_cpus();
// the quota alone, on any cgroup tree (the tests hand in a fake one): whole CPUs, 0 = no quota found
unsigned cgroup_cpu_quota(const std::string& sysfs_cgroup_root, const std::string& proc_self_cgroup);

// Threads that are joined whatever path leaves the scope.  A std::thread constructor that throws (EAGAIN) halfway
// through a pool would otherwise destroy joinable threads: std::terminate, which no catch at the C boundary can stop
// (ADVICE r3).  The workers must end by themselves (they run to completion or watch a flag the owner sets).
// An exception INSIDE a worker (an allocation that fails while a directory is listed, a record written) would end the
// process the same way -- no catch at the C boundary is on that thread's stack: spawn() runs the worker under a catch,
// remembers that it threw, and join_all() raises std::bad_alloc on the owner's thread, where the entry point's catch is.
struct ThreadJoiner {
    std::vector<std::thread> th;
    std::atomic<bool> worker_threw{false};
    ThreadJoiner() = default;
    ThreadJoiner(const ThreadJoiner&) = delete;
    ThreadJoiner& operator=(const ThreadJoiner&) = delete;
    ~ThreadJoiner() { join_quietly(); }
    template <class F, class... A> void spawn(F&& f, A&&... a)
    {
        th.emplace_back([this](auto fn, auto... args) {
            try { fn(args...); } catch (...) { worker_threw.store(true); }
        }, std::forward<F>(f), std::forward<A>(a)...);
    }
    void join_quietly()
    {
        for (auto& t : th)
            if (t.joinable()) t.join();
    }
    void join_all()
    {
        join_quietly();
        if (worker_threw.exchange(false)) throw std::bad_alloc();
    }
};

// ---- a persistent pool of fill threads, pinned to a CPU set ---------------------------------------------------
// parallel_for(n, T, fn): fn(i) for i in [0, n) on at most T of the pool's threads plus the caller; returns when
// all are done.  One call at a time (an engine has one batch being filled at any moment).
class FillPool {
public:
    FillPool() = default;
    ~FillPool();
    FillPool(const FillPool&) = delete;
    FillPool& operator=(const FillPool&) = delete;
    // (re)sizes the pool; threads are created lazily by parallel_for.  cpus empty = no affinity.
    void configure(unsigned max_threads, const std::vector<int>& cpus);
    void parallel_for(size_t n, unsigned threads, const std::function<void(size_t)>& fn);
    unsigned max_threads() const { return max_threads_; }
    const std::vector<int>& cpus() const { return cpus_; }

private:
    void worker(unsigned id);
    void stop();
    unsigned max_threads_ = 0;
    std::vector<int> cpus_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    uint64_t epoch_ = 0;       // bumped per parallel_for
    unsigned want_ = 0;        // pool threads asked to join the current job
    unsigned running_ = 0;     // pool threads still inside the current job
    bool quit_ = false;
    const std::function<void(size_t)>* fn_ = nullptr;
    std::atomic<bool> fn_threw_{false}; // a pool thread's fn(i) threw: parallel_for raises std::bad_alloc on the caller's thread
    size_t n_ = 0;
    std::atomic<size_t>* next_ = nullptr;
};

// fn(t) for t in [0, T), every t on a thread of its own (the caller runs t = 0): on the process-wide pool of host helper
// threads when nobody else is using it (round 5: the walk, the YAML writer and the parser started 15 fresh threads a
// phase, ~25 us each, serial on the caller -- 0.6 ms of a 2.8 ms walk), on fresh threads otherwise (two contexts walking
// at once).  Raises std::bad_alloc on the caller's thread when a worker threw, like ThreadJoiner::join_all.
void run_on_threads(unsigned T, const std::function<void(unsigned)>& fn);

// memcpy into a staging buffer that the CPU will not read again: non-temporal stores for large pieces (no
// read-for-ownership of the destination lines: a quarter less host memory traffic per staged byte), plain memcpy
// for small ones.
void copy_to_staging(void* dst, const void* src, size_t n);

} // namespace snaphash
