// member_hashers.h -- the fused Build pass's LONG members, hashed on host threads where the packer put their bytes.
// Internal, host-only (no HIP): tests/tsan_host.cpp drives it under ThreadSanitizer.  Reference: the two SHA-512 passes of
// `snappy build` (snappy/build.go:222, :241 through helpers.Sha512sum, helpers/helpers.go:187-201) read every file a second
// time on one goroutine; here a long member's digest comes from the bytes the tar packer has just read.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "hostsha.h"

namespace snaphash {

// Every member's SHA-512 used to go to the kernels, out of the staged tar stream -- and a lone SHA-512 chain advances
// at 44 MB/s on the GPU: a package of 3 000 files with a 16 MiB binary among them took 460 ms where
// its bytes pass in 80, a single 1 MiB file 24 ms (tools/build_small_probe.py); and a hashing workgroup beside the
// compressor costs it a fifth round of workgroups (profiles/r05_deflate_depths.txt).  Which members come here is the
// producer's decision (targz.inc: all of them with eight cores or more, the long ones with fewer, none under
// SNAPHASH_FLAG_GPU_ONLY); this is the mechanism: a member's bytes are hashed FROM THE PINNED STAGING BUFFER the packer has
// just read them into -- every file is still read once -- its pieces (it may span slots) go to ONE worker in slot order,
// and a slot's host buffer is not refilled while a worker still reads it.
struct MemberHashers {
    struct Task { uint32_t h; const uint8_t* p; uint64_t n; bool first, fin; int slot; };
    struct Worker { std::thread th; std::deque<Task> q; };
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<HostSha> chain;       // per hosted member
    std::vector<uint32_t> worker_of;  // per hosted member
    std::vector<uint8_t> digests;     // per hosted member, 64 bytes
    int outstanding[2] = {0, 0};      // tasks still reading slot b's host buffer
    bool quit = false;
    uint64_t bytes = 0;
    ~MemberHashers() { stop(); }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_work.notify_all();
        for (auto& w : workers)
            if (w->th.joinable()) w->th.join();
        workers.clear();
    }
    // sizes of the hosted members, in the order their ids are given out
    void start(const std::vector<uint64_t>& sizes, unsigned nworkers)
    {
        const size_t n = sizes.size();
        chain.resize(n);
        digests.assign(n * 64, 0);
        worker_of.assign(n, 0);
        nworkers = (unsigned)std::min<size_t>(std::max(1u, nworkers), n);
        std::vector<uint64_t> load(nworkers, 0); // longest first onto the least loaded worker
        std::vector<uint32_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sizes[a] > sizes[b]; });
        for (uint32_t h : order) {
            const unsigned w = (unsigned)(std::min_element(load.begin(), load.end()) - load.begin());
            worker_of[h] = w;
            load[w] += sizes[h];
            bytes += sizes[h];
        }
        for (unsigned w = 0; w < nworkers; ++w) workers.emplace_back(new Worker());
        for (unsigned w = 0; w < nworkers; ++w) workers[w]->th = std::thread([this, w] { run(w); });
    }
    void run(unsigned w)
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || !workers[w]->q.empty(); });
                if (workers[w]->q.empty()) return; // (quit: what is queued is still done first)
                t = workers[w]->q.front();
                workers[w]->q.pop_front();
            }
            if (t.first) host_sha512_init(chain[t.h]);
            host_sha512_update(chain[t.h], t.p, t.n);
            if (t.fin) host_sha512_final(chain[t.h], digests.data() + 64 * (size_t)t.h);
            {
                std::lock_guard<std::mutex> lk(mu);
                --outstanding[t.slot];
            }
            cv_done.notify_all();
        }
    }
    void give(const Task& t)
    {
        { std::lock_guard<std::mutex> lk(mu); ++outstanding[t.slot]; workers[worker_of[t.h]]->q.push_back(t); }
        cv_work.notify_all();
    }
    void give_all(const std::vector<Task>& ts) // a slot's worth at once: one lock, one wake-up
    {
        if (ts.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (const Task& t : ts) { ++outstanding[t.slot]; workers[worker_of[t.h]]->q.push_back(t); }
        }
        cv_work.notify_all();
    }
    void wait_slot(int slot) // before the slot's host buffer is written again
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return outstanding[slot] == 0; });
    }
    void wait_all()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return outstanding[0] == 0 && outstanding[1] == 0; });
    }
};

} // namespace snaphash
