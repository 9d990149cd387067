// planner.cpp -- see planner.h.  Host-only (no HIP): the CPU suite drives it through snaphash_plan_streams.
#include "planner.h"

#include <algorithm>
#include <functional>
#include <queue>

namespace snaphash {

namespace {

// as snaphash_api.cpp plan_kernels decides (the lane-pair kernel takes few, long streams; a heavy-tailed batch is cut
// into a long head for it and a short tail for the lane-per-stream kernel)
constexpr size_t kPairMaxStreams = 32768;
constexpr uint64_t kPairMinBlocks = 32;
constexpr uint64_t kMinHostFile = 256u << 10; // a file smaller than this does not move to a host thread while there is a GPU part (below)
constexpr double kThreadWorth = 250e-6; // a further host thread is started per this much host work (a start costs ~30 us)

uint64_t blocks_of(uint64_t len) { return (len >> 7) + 1; }

using MinHeap = std::priority_queue<double, std::vector<double>, std::greater<double>>;

// order[] = the streams longest first, equal lengths in list order (what std::stable_sort by length gives -- and took 6 of
// the 14 ms config 5's 100 000 streams were planned in: a merge sort through an index, every compare two cache misses).
// Large lists: a radix sort over the bits the longest length has, 11 a pass, least significant first -- stable by
// construction, three passes for anything under 8 GiB.
void longest_first(const uint64_t* lens, size_t n, std::vector<uint32_t>& order)
{
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    if (n < 4096) {
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lens[a] > lens[b]; });
        return;
    }
    uint64_t longest = 0;
    for (size_t i = 0; i < n; ++i) longest = std::max(longest, lens[i]);
    unsigned bits = 0;
    while (bits < 64 && (longest >> bits) != 0) ++bits;
    constexpr unsigned kDigit = 11;
    std::vector<uint32_t> other(n);
    std::vector<uint64_t> key(n), key_other(n); // longest - len: ascending keys = descending lengths
    for (size_t i = 0; i < n; ++i) key[i] = longest - lens[i];
    for (unsigned shift = 0; shift < bits; shift += kDigit) {
        size_t count[(1u << kDigit) + 1] = {0};
        for (size_t i = 0; i < n; ++i) ++count[((key[i] >> shift) & ((1u << kDigit) - 1u)) + 1u];
        for (size_t d = 0; d < (1u << kDigit); ++d) count[d + 1] += count[d];
        for (size_t i = 0; i < n; ++i) {
            const size_t at = count[(key[i] >> shift) & ((1u << kDigit) - 1u)]++;
            other[at] = order[i];
            key_other[at] = key[i];
        }
        order.swap(other);
        key.swap(key_other);
    }
}

} // namespace

namespace {
// the first observation stands, later ones move the estimate a quarter of the way (one odd batch does not re-plan a box)
bool take(double& est, unsigned& count, double rate, double lo, double hi)
{
    if (!(rate >= lo && rate <= hi)) return false;
    est = count == 0 ? rate : 0.75 * est + 0.25 * rate;
    ++count;
    return true;
}
// what the model calls the link is the staged pass end to end, a little under the copies' own rate: 54 GB/s of files and
// 55 of memory where the events say 56.7 (profiles/r04_h2d_probe.txt)
constexpr double kLinkOfDmaFiles = 54.0 / 56.7, kLinkOfDmaMem = 55.0 / 56.7;
} // namespace

// An observation enters the estimate already cut to what apply() would believe of it (below), so that one absurd call --
// the first read of freshly written tmpfs files measured 0.21 GB/s a thread -- costs the estimate one step, not eight.
bool PlanCalib::observe_dma(double bytes, double seconds)
{
    if (bytes < (double)(4u << 20) || seconds < 50e-6) return false;
    const double rate = bytes / seconds;
    if (!(rate >= 2e9 && rate <= 400e9)) return false;
    return take(dma, n_dma, std::min(4.0 * 56.7e9, std::max(0.25 * 56.7e9, rate)), 2e9, 400e9);
}

bool PlanCalib::observe_fill(bool files, double bytes, double thread_seconds)
{
    if (bytes < (double)(4u << 20) || thread_seconds < 50e-6) return false;
    const double rate = bytes / thread_seconds, dflt = files ? 6.5e9 : 9e9;
    if (!(rate >= 0.05e9 && rate <= 200e9)) return false;
    const double cut = std::min(2.0 * dflt, std::max(0.5 * dflt, rate));
    return files ? take(fill_files, n_fill_files, cut, 0.05e9, 200e9) : take(fill_mem, n_fill_mem, cut, 0.05e9, 200e9);
}

bool PlanCalib::observe_fill_per_file(double files, double thread_seconds)
{
    if (files < 256 || thread_seconds <= 0) return false;
    const double each = thread_seconds / files;
    if (!(each >= 0.05e-6 && each <= 1e-3)) return false;
    return take(fill_per_file, n_fill_per_file, std::min(2.0 * 10e-6, std::max(0.3 * 10e-6, each)), 0.05e-6, 1e-3);
}

void PlanCalib::observe_call(bool files, double bytes, double streams, double copies, double h2d_seconds, double fill_thread_seconds, bool take_fill)
{
    if (copies >= 1 && bytes / copies >= (double)(32u << 20)) observe_dma(bytes, h2d_seconds);
    if (!take_fill || streams < 1 || bytes <= 0) return;
    const double each = files ? (fill_per_file > 0 ? fill_per_file : 10e-6) : 0.3e-6; // what the model charges a stream
    const double rate = files ? (fill_files > 0 ? fill_files : 6.5e9) : (fill_mem > 0 ? fill_mem : 9e9);
    const double mean = bytes / streams;
    if (mean >= (double)(256u << 10)) {
        const double net = fill_thread_seconds - streams * each;
        if (net > 0.5 * fill_thread_seconds) observe_fill(files, bytes, net);
    } else if (files && mean < (double)(64u << 10)) {
        const double net = fill_thread_seconds - bytes / rate;
        if (net > 0.5 * fill_thread_seconds) observe_fill_per_file(streams, net);
    }
}

bool PlanCalib::observe_host(double planned_s, double actual_s)
{
    if (planned_s < 5e-3 || actual_s < 5e-3) return false;
    const double g = host_gain > 0 ? host_gain : 1.0;
    const double seen = g * planned_s / actual_s; // the gain that would have made the plan come true
    if (!(seen > 0.1 && seen < 10.0)) return false;
    return take(host_gain, n_host, std::min(1.6, std::max(0.6, seen)), 0.1, 10.0);
}

void PlanCalib::relax(bool files)
{
    double& est = files ? fill_files : fill_mem;
    const double dflt = files ? 6.5e9 : 9e9;
    if (est > 0 && est < dflt) est = 0.75 * std::max(est, 0.5 * dflt) + 0.25 * dflt;
}

// What the observations may do to the model.  The link: anything a PCIe generation or two away from the 56.7 GB/s the
// defaults were taken on.  A fill thread: DOWN to half of the default, never up -- the default (9 / 6.5 GB/s) is not what
// a thread can move (a thread alone copies 40 GB/s out of DRAM on the box the defaults come from, and 12-14 inside the
// pipeline: profiles/r05_calibration.txt) but what makes the model's "cores" bound come out right beside host threads
// that compete with it; a box whose fills are slower scales it down, a faster one leaves it alone.  And an observation is
// not believed beyond that range whatever it says: the first read of freshly written tmpfs files measured 0.22 GB/s a
// thread, once, and a model that believed it would have sent every later call to the host threads for good.
void PlanCalib::apply(PlanModel& m) const
{
    if (m.gpu_link <= 0 && dma > 0)
        m.gpu_link = std::min(4.0 * 56.7e9, std::max(0.25 * 56.7e9, dma)) * (m.from_files ? kLinkOfDmaFiles : kLinkOfDmaMem);
    if (host_gain > 0) m.host_rate *= std::min(1.6, std::max(0.6, host_gain));
    if (m.fill_rate <= 0) {
        const double dflt = m.from_files ? 6.5e9 : 9e9;
        const double seen = m.from_files ? fill_files : fill_mem;
        if (seen > 0) m.fill_rate = std::min(dflt, std::max(0.5 * dflt, seen));
    }
    // what a file costs a fill thread: both ways (the descriptor table's lock is a box's own: its cores, its kernel)
    if (m.fill_per_stream <= 0 && m.from_files && fill_per_file > 0) m.fill_per_stream = std::min(2.0 * 10e-6, std::max(0.3 * 10e-6, fill_per_file));
}

PlanResult plan_streams(const uint64_t* lens, size_t n, const PlanModel& m)
{
    PlanResult res;
    res.on_host.assign(n, 0);
    if (n == 0) return res;
    const unsigned nd = std::max(1u, m.n_devices);
    const double link = (m.gpu_link > 0 ? m.gpu_link : (m.from_files ? 54e9 : 55e9)) * nd;
    const double g_plan = m.gpu_per_stream > 0 ? m.gpu_per_stream : 0.15e-6;
    const double f_rate = m.fill_rate > 0 ? m.fill_rate : (m.from_files ? 6.5e9 : 9e9);
    const double f_stream = m.fill_per_stream > 0 ? m.fill_per_stream : (m.from_files ? 10e-6 : 0.3e-6);
    const double h_rate = m.host_rate > 0 ? m.host_rate : 1.4e9;
    const unsigned cpus = std::max(1u, m.cpus);
    const unsigned fill = std::max(1u, m.fill_threads) * nd;
    const unsigned h_base = cpus > fill + 1u ? cpus - fill - 1u : 1u; // beside a GPU part at full fill: its fill threads and the engine's own thread keep their cores
    const unsigned h_alone = m.host_threads ? m.host_threads : cpus;   // no GPU part: every core hashes
    // a stream's fixed cost on a host thread: for files open + close, which do not scale -- every thread of a process takes
    // the lock of its ONE descriptor table (openat relative to the directory's descriptor changes nothing, lstat scales
    // fine: profiles/r04_openat_probe.txt): 3 us alone, 19 us each with twelve at it, 25 with sixteen (100 000 files)
    auto h_stream = [&](unsigned threads) {
        if (m.host_per_stream > 0) return m.host_per_stream;
        return m.from_files ? 4e-6 * std::max(1.0, (double)threads / 3.0) : 0.05e-6;
    };

    std::vector<uint32_t> order(n);
    longest_first(lens, n, order);
    std::vector<double> suffix(n + 1, 0.0), suffix_blocks(n + 1, 0.0);
    for (size_t k = n; k-- > 0;) {
        suffix[k] = suffix[k + 1] + (double)lens[order[k]];
        suffix_blocks[k] = suffix_blocks[k + 1] + (double)blocks_of(lens[order[k]]);
    }
    // CPU time the staging fill of streams order[k..] costs (all fill threads together)
    auto fill_work = [&](size_t k) { return k >= n ? 0.0 : suffix[k] / f_rate + (double)(n - k) * f_stream; };
    // modelled time of the GPU part when streams order[k..] stay on it
    auto gpu_time = [&](size_t k) {
        if (k >= n) return 0.0;
        const size_t left = n - k;
        const uint64_t b0 = blocks_of(lens[order[k]]);
        bool pair;
        if (left <= kPairMaxStreams) pair = suffix_blocks[k] >= (double)kPairMinBlocks * (double)left;
        else pair = b0 >= kPairMinBlocks && b0 >= 8 * blocks_of(lens[order[k + kPairMaxStreams - 1]]);
        const double rate = pair ? m.gpu_pair_rate : m.gpu_wide_rate;
        return m.gpu_latency + (double)left * g_plan + std::max({(double)lens[order[k]] / rate, suffix[k] / link, fill_work(k) / (double)fill});
    };

    // The k longest streams on h host threads (LPT), the rest on the GPU: G falls and H rises with k.  The fill threads
    // keep their count whatever h is -- they are busy in bursts, and taking cores from them cost more than it gave
    // (214 ms against 179 on config 2, profiles/r04_default_probe.txt) -- so h + fill may exceed the cores; what binds
    // then is the cores' total: (fill work + host work) / cpus.
    const double gpu_alone = gpu_time(0);
    double best = gpu_alone, best_host = 0;
    size_t best_k = 0;
    unsigned threads = h_base;
    const unsigned h_last = m.host_threads ? m.host_threads : (cpus > 2u ? std::max(h_base, cpus - 2u) : h_base);
    for (unsigned h = m.host_threads ? m.host_threads : h_base; h <= h_last; ++h) {
        const double hs = h_stream(h);
        MinHeap pool;
        for (unsigned t = 0; t < h; ++t) pool.push(0.0);
        double host_makespan = 0, host_work = 0, host_bytes = 0;
        const double lane_gain = std::max(1.0, m.host_lane_gain);
        for (size_t k = 0; k < n; ++k) {
            // A small file costs its open + close more than its bytes, and that cost is the same lock whoever pays it (the
            // process has one descriptor table): beside a GPU part it stays with the fill threads -- 100 000 x 8 KiB took
            // 148 ms with a quarter of them on host threads against 115 ms whole (profiles/r04_small_files_tree.txt).
            if (m.from_files && lens[order[k]] < kMinHostFile) break;
            // a stream long against its thread's share keeps the core to itself, the others run eight at a time
            // (snaphash_api.cpp run_host; the share here is what has moved so far: the long ones move first)
            host_bytes += (double)lens[order[k]];
            const bool lanes = lane_gain > 1.0 && k + 1 >= 3u * h && (double)lens[order[k]] < host_bytes / h / 4.0;
            const double c = (double)lens[order[k]] / (lanes ? h_rate * lane_gain : h_rate) + hs;
            const double t = pool.top() + c;
            pool.pop();
            pool.push(t);
            host_work += c;
            host_makespan = std::max(host_makespan, t);
            const double g = gpu_time(k + 1);
            const double cores = k + 1 < n ? (fill_work(k + 1) + host_work + g) / (double)cpus : 0.0; // (+ g: the engine's own thread)
            const double mk = std::max({g, host_makespan, cores});
            if (mk < best * 0.98) { best = mk; best_k = k + 1; best_host = host_makespan; threads = h; } // move only for a real gain
            if (host_makespan > best) break;                                                           // H only grows from here
            // ... and once the host threads are what the split waits for, every further stream they take only makes it wait
            // longer (H never falls).  Without this a job with one dominant stream -- config 5: its 255 MiB head alone is the
            // makespan from k = 1 on -- walked all 100 000 streams for every thread count: 14 ms of planning in front of a
            // 190 ms call (profiles/r05_bench_C2_default.json configs.C5), now 33 streams a thread count.
            if (host_makespan >= g && host_makespan >= cores) break;
        }
    }
    // A GPU part that is bound by its LINK (many similar streams: no stream dominates) gains from the host only what
    // the spare cores add to the link's rate, and those cores share memory bandwidth and the CPU quota with the fill threads
    // and the copy engine: measured on the GPU box, a modelled 5-8 % came out between +2 % and -7 %
    // (profiles/r04_default_probe.txt).  Such a batch is split only for a modelled 10 % or more.
    if (best_k > 0 && best_k < n && suffix[0] / link >= (double)lens[order[0]] / m.gpu_pair_rate && best > 0.90 * gpu_alone) {
        best = gpu_alone;
        best_k = 0;
        best_host = 0;
    }
    // no GPU part at all: the fill threads' cores hash too
    if (best_k < n || h_alone > threads) {
        const double hs = h_stream(h_alone);
        // (every stream's share is known here: who runs eight at a time is the run-time rule itself)
        const double lane_gain = n >= 3u * h_alone ? std::max(1.0, m.host_lane_gain) : 1.0;
        const double alone_from = suffix[0] / h_alone / 4.0;
        auto cost = [&](size_t k) { return (double)lens[order[k]] / ((double)lens[order[k]] < alone_from ? h_rate * lane_gain : h_rate) + hs; };
        double work = 0;
        for (size_t k = 0; k < n; ++k) work += cost(k);
        if (std::max(work / h_alone, cost(0)) < best * 0.98) { // the lower bound first: the LPT pass is O(n log threads)
            MinHeap pool;
            for (unsigned t = 0; t < h_alone; ++t) pool.push(0.0);
            double mk = 0;
            for (size_t k = 0; k < n && mk < best; ++k) {
                const double t = pool.top() + cost(k);
                pool.pop();
                pool.push(t);
                mk = std::max(mk, t);
            }
            if (mk < best * 0.98) { best = mk; best_k = n; best_host = mk; threads = h_alone; }
        }
    }

    double work = 0;
    const double hs_used = h_stream(threads);
    for (size_t k = 0; k < best_k; ++k) {
        res.on_host[order[k]] = 1;
        res.host_bytes += lens[order[k]];
        work += (double)lens[order[k]] / h_rate + hs_used;
    }
    res.host_streams = best_k;
    res.gpu_seconds = gpu_time(best_k);
    res.host_seconds = best_host;
    if (best_k) {
        const double want = work / kThreadWorth + 1.0;
        res.host_threads = (unsigned)std::min<double>({(double)threads, (double)best_k, want});
        if (res.host_threads == 0) res.host_threads = 1;
    }
    return res;
}

} // namespace snaphash
