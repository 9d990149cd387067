// planner.h -- which streams of a call the HIP kernels hash and which the library's own host SHA-512 (hostsha.cpp).
// Internal; snaphash_plan_streams (include/snaphash.h) exposes it for the tests and for callers who want to see a plan.
//
// No reference counterpart: helpers.Sha512sum (helpers/helpers.go:187-201) is one goroutine.  The seam it serves is
// that function and its loop in writeHashes (snappy/build.go:222, :241): a call through this library must never be
// slower than the loop it replaces, and ONE SHA-512 stream advances at ~44 MB/s on MI355X (the chain is serial;
// DESIGN.md sec. 4) against ~1.4 GB/s on a host core -- so the GPU is the right place for many streams at once and the
// wrong one for a lone file, a tree dominated by one member, or the package's own data.tar.gz.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace snaphash {

struct PlanModel {
    unsigned n_devices = 1;       // engines the GPU part is sharded over
    unsigned cpus = 1;            // host cores the call may keep busy (affinity and cgroup quota, hostfill.h usable_cpus)
    unsigned fill_threads = 6;    // cores ONE engine's staging fill occupies while there is a GPU part
    unsigned host_threads = 0;    // 0 = automatic (cpus, less the fill threads while there is a GPU part); N = exactly N
    bool from_files = false;      // sources are paths (open + pread + close per stream) rather than caller memory
    double host_rate = 1.4e9;     // B/s of one host core's SHA-512 (measured at snaphash_init)
    double host_per_stream = 0;   // s per stream on a host thread; 0 = 4 us for files, 0.05 us for memory
    double gpu_pair_rate = 44e6;  // B/s of ONE stream under the lane-pair kernel (few, long streams)
    double gpu_wide_rate = 18e6;  // B/s of ONE stream under the lane-per-stream kernel
    double gpu_link = 0;          // B/s one engine stages and copies (PCIe inclusive); 0 = 55e9 memory, 54e9 files (measured end to end, r04)
    double gpu_latency = 150e-6;  // s a launch costs whatever its size: job upload, kernel start, sync, digests back
    double gpu_per_stream = 0;    // s per stream of batch planning on the engine's thread; 0 = 0.15 us
    double fill_rate = 0;         // B/s ONE fill thread moves into the staging buffers; 0 = 9e9 memory, 6.5e9 files (pread)
    double fill_per_stream = 0;   // s per stream on a fill thread (files: open + close beside eleven others); 0 = 0.3 us memory, 10 us files
    double host_lane_gain = 1.0;  // what a host thread gains from running its streams eight at a time (hostsha_x8.cpp): x the one-stream rate, for
                                  // streams short enough to share a core (under a quarter of a thread's share of the host part)
};

// What THIS box does, as far as the library has seen it (round 5: the model's constants were one lease's 16-CPU quota and
// one PCIe link; the 8-GPU node is another box).  The two constants that move most between boxes -- what an engine's
// copy engine moves over its link and what one fill thread moves into the staging buffers -- are measured: the link
// when a ctx that may plan is created (two timed copies, ~1 ms) and by every staged call afterwards (HIP events of
// the H2D copies), the fill threads by every staged call but a ctx's first (wall x threads of its fills);
// snaphash_stats_ex sets the model's prediction beside what the call took.  apply() says how far an observation is believed.
// Host-only arithmetic: the CPU suite drives it through snaphash_calib_observe.
struct PlanCalib {
    double dma = 0;         // B/s one engine's H2D copies run at (HIP events); 0 = not measured: the model's default link
    double fill_mem = 0;    // B/s ONE fill thread copies from caller memory into pinned staging
    double fill_files = 0;  // B/s ONE fill thread preads from the page cache into pinned staging
    double host_gain = 0;   // what a host thread REALLY did over what the model said it would (both with the gain in force then): x the model's
                            // host rate; 0 = not measured (1.0).  The lane gain of the eight-stream hasher (2.4 files / 3.2 memory) is one box's
                            // number and conservative there: the C2 tree's host part was planned at 147 ms and took 130
    double fill_per_file = 0; // s a FILE costs a fill thread whatever its length (open + close beside the other threads'); 0 = not measured (10 us)
    unsigned n_dma = 0, n_fill_mem = 0, n_fill_files = 0, n_host = 0, n_fill_per_file = 0; // observations taken
    // an observation: bytes moved in `seconds` (link: summed event time of the copies; fill: wall x threads at it).
    // Too small to mean anything (under 4 MiB, under 50 us) or outside what any box does: ignored (returns false).
    bool observe_dma(double bytes, double seconds);
    bool observe_fill(bool files, double bytes, double thread_seconds);
    // `files` files cost the fill threads `thread_seconds` beyond what their bytes took: 256 files or more, 0.3 .. 2 x the default 10 us
    bool observe_fill_per_file(double files, double thread_seconds);
    // What ONE staged call of an engine says, sorted into the observations above by what it can speak about (round 5: a tree of
    // 5 000 x 8 KiB moved the LINK estimate to 41 GB/s -- three copies of 12 MiB measure their latency -- and the fill rate to
    // its floor -- the threads spent their time in open(), not in bytes -- and the next big tree was planned with both):
    // the link from copies of 32 MiB and more on average, the fill rate from streams of 256 KiB and more on average and net
    // of the per-file cost, the per-file cost from calls of small files (under 64 KiB on average) net of their bytes.
    void observe_call(bool files, double bytes, double streams, double copies, double h2d_seconds, double fill_thread_seconds, bool take_fill);
    // a host part that was planned at planned_s took actual_s (busiest thread): parts under 5 ms say nothing
    bool observe_host(double planned_s, double actual_s);
    // A call that was planned onto host threads whole measured no fill: a low estimate that caused that plan would never be
    // corrected.  Such a call moves the estimate a quarter of the way back to the model's default, so that the GPU part is
    // tried again -- and measured again -- after a few of them.
    void relax(bool files);
    void apply(PlanModel& m) const; // fills gpu_link / fill_rate of a model that has not set them
};

struct PlanResult {
    std::vector<uint8_t> on_host; // per stream: 1 = a host thread hashes it
    unsigned host_threads = 0;    // threads the host part should run on (0 = no host part)
    double gpu_seconds = 0;       // modelled makespan of the GPU part (0 = no GPU part)
    double host_seconds = 0;      // modelled makespan of the host part
    uint64_t host_streams = 0, host_bytes = 0;
};

// The GPU part costs its launch latency + per-stream planning + the largest of: its longest stream at the kernel's
// per-stream rate, its bytes over the link, its fill work over its fill threads.  The host part is LPT over its
// threads.  With host_threads = 0 every host thread count from "what the fill threads and the engine leave" up to
// "all cores but two" is tried, the fill threads keeping theirs (they work in bursts); beyond the cores' number what
// binds is their total: (fill work + host work) / cpus.  A GPU part bound by its longest stream (a tree of many small
// files and a few big ones) leaves cores to the big files; one bound by its link gives up the share the cores can hash
// while they also feed it.
// Streams move to the host longest first -- the longest sets the GPU's makespan and costs the host least per byte of
// relief -- while that shortens max(GPU, host); and the whole batch moves when the host alone beats every split (a
// batch too small to repay a launch, or one whose longest member is most of it).
PlanResult plan_streams(const uint64_t* lens, size_t n, const PlanModel& m);

} // namespace snaphash
