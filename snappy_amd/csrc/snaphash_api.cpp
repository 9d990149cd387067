// snaphash_api.cpp -- the C ABI of libsnaphash.so (include/snaphash.h): context,
// HBM/pinned staging, the staging engine (batches in sub-slots, fill pool, copy stream)
// that feeds the multi-buffer SHA-512 kernels, the plan of a call (planner.h), the in-library
// and per-process multi-GPU forms, and the writeHashes / Verify passes built on them.
//
// Reference behaviour mirrored here (paths relative to the upstream tree):
//   helpers/helpers.go:187-201  Sha512sum: whole-file digest, any read error fails
//   snappy/build.go:216-270     writeHashes: archive digest first, then the walk;
//                               the first error aborts, nothing is written
// There is no fallback in this file: without a gfx950 device nothing runs, and no error re-routes a call.  Which streams
// of a call the kernels take and which the library's own host SHA-512 (hostsha.cpp) is planned before anything runs
// (hash_sources_top) and reported afterwards (snaphash_stats_ex); SNAPHASH_FLAG_GPU_ONLY plans nothing.
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <queue>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h> // types only: the library is dlopen'ed when a ctx owns several devices

#include <condition_variable>
#include <future>
#include <memory>
#include <mutex>

#include "deflate_kernels.h"
#include "hostfill.h"
#include "hostpass.h"
#include "hostsha.h"
#include "member_hashers.h"
#include "planner.h"
#include "sha512_core.h"
#include "sha512_kernels.h"
#include "tarpack.h"
#include "walk.h"

using namespace snaphash;

namespace {

constexpr uint64_t kDefaultStaging = 256ull << 20;
constexpr uint64_t kAlign = 256;          // placement of a segment inside a staging buffer
constexpr uint64_t kMinSegment = 32u << 10;    // least a FILE stream is given of a slot, unless it ends there (a pread per segment; an open + close
                                               // too once the call holds more descriptors than its budget, FdCache)
constexpr uint64_t kMinSegmentMem = 16u << 10; // the same for a stream in caller memory (a copy has no such cost)
constexpr uint32_t kTargetStreams = 4096; // streams per batch the engine aims for (keeps the kernel ahead of PCIe)

struct EventPair { hipEvent_t a = nullptr, b = nullptr; int kind = 0; }; // kind 0 SHA-512 kernels, 1 h2d, 2 deflate kernels

struct Slot {
    uint8_t* h_buf = nullptr; // pinned host
    uint8_t* d_buf = nullptr; // HBM
    uint64_t cap = 0;         // bytes both hold (the engine's staging size, or less while only small jobs have come by)
    Job* h_jobs = nullptr;    // pinned
    Job* d_jobs = nullptr;
    size_t jobs_cap = 0;
    hipEvent_t done = nullptr;   // kernel of the batch staged in this slot has finished
    hipEvent_t copied = nullptr; // H2D of this slot's data + jobs has finished
    bool busy = false;
};

// A batch of the hashing engine lives in a sub-slot: a piece of one of the engine's staging buffers with its own job
// array and events.  A job of several buffers' worth is cut into batches much smaller than a buffer (hash_sources):
// when the streams' own rate is about the link's (a rank's shard of config 4: 1 250 streams x 44 MB/s = 55 GB/s), fill,
// copy and kernel are three stages of equal length and only many small batches in flight keep all three busy.
struct SubSlot {
    Job* h_jobs = nullptr; // pinned
    Job* d_jobs = nullptr;
    size_t jobs_cap = 0;
    hipEvent_t done = nullptr;   // kernel of the batch staged here has finished
    hipEvent_t copied = nullptr; // H2D of this batch's jobs has finished
    bool busy = false;
};

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct snaphash_ctx;

// One engine: a device, its streams, staging buffers and kernel scratch.  A snaphash_ctx owns one
// or several of these.
struct DevCtx {
    int device = 0;
    int index = 0; // position in snaphash_ctx::dev
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr; // H2D of batch k+1 overlaps the kernel of batch k
    uint64_t staging = kDefaultStaging;
    uint32_t kernel_pref = SNAPHASH_KERNEL_AUTO;
    uint32_t deflate_depth = 0; // hash-chain links the DEFLATE search walks per position (snaphash_config.deflate_depth; 0 = kDfDepth)
    uint32_t n_xcd = 8; // XCDs (L2 domains) the device presents: 8 in SPX mode; the DEFLATE launch keeps a run of chunks on one

    std::vector<SubSlot> sub; // the hashing engine's batches in flight (pieces of slot[0..2])
    Slot slot[3]; // [0], [1]: every staging user; [2]: a third buffer for the hashing engine alone, allocated when a call
                  // has more than two buffers' worth of bytes (fill k+2 then overlaps kernel k: with two, a job whose
                  // kernels take as long as its copies -- one rank's shard of config 4 -- idles between batches)
    // device-resident entry point
    Job* h_jobs = nullptr; // pinned
    Job* d_jobs = nullptr;
    size_t jobs_cap = 0;
    uint64_t* d_state = nullptr;
    size_t state_cap = 0; // streams
    uint8_t* d_digests = nullptr;
    size_t digests_cap = 0; // streams

    CmpChunk* h_chunks = nullptr; // pinned (range comparison)
    CmpChunk* d_chunks = nullptr;
    size_t chunks_cap = 0;
    uint8_t* d_equal = nullptr;
    size_t equal_cap = 0;

    // block-parallel DEFLATE scratch (row f3): per-chunk output slots, sizes, offsets, compacted output
    uint8_t* d_zslots = nullptr;
    uint8_t* d_zout = nullptr;
    uint32_t* d_ztoks = nullptr; // the deflate kernel's parse, one word per staged byte
    uint32_t* d_zsizes = nullptr;
    uint64_t* d_zprefix = nullptr;
    uint32_t* h_zsizes = nullptr; // pinned
    uint64_t* h_zprefix = nullptr;
    uint8_t* h_zout[2] = {nullptr, nullptr}; // pinned, double-buffered: the consumers read one while the next D2H fills the other
    hipStream_t z_stream = nullptr;          // the compressor's stream
    hipStream_t z2_stream = nullptr;         // concatenation of a finished piece and its way back to the host
    std::vector<hipEvent_t> z_ev;            // "the sizes of piece k are on the host"
    std::vector<hipEvent_t> z_part_ev;       // "part k of the pass's first slot is in HBM" (targz.inc)
    size_t z_chunks = 0;

    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    bool pending = false;

    // host side of the staging (hostfill.h): the GPU's NUMA node, the pool of fill threads that lives there
    std::string pci_bus_id;
    int numa_node = -1;        // -1: unknown or not applied (single-node host, SNAPHASH_FLAG_NO_NUMA)
    int staging_node = -1;     // node the first staging page was found on after allocation (diagnostic)
    int64_t fd_budget = 0;     // file descriptors a hashing call may keep open between batches (FdCache)
    int64_t fd_call_budget = -1; // fewer than that for the call in progress (its host lanes hold descriptors too; -1 = no)
    unsigned fill_cap = 12;    // most fill threads this engine uses (the ctx divides the usable CPUs among its engines)
    unsigned fill_call_cap = 0; // fewer than that for the call in progress (a rank's share of the node's cores; 0 = no)
    snaphash_ctx* owner = nullptr; // for what the engine measures about the box (planner.h PlanCalib)
    unsigned owner_cpus_per_engine = 0; // usable CPUs over the ctx's engines
    uint64_t staged_calls = 0;  // hash_sources calls completed on this engine
    double fill_thread_s = 0;   // this call: wall x threads of its staging fills ...
    uint64_t fill_bytes = 0;    // ... and the bytes they moved
    FillPool pool;

    snaphash_stats stats{};
    double t_call0 = 0;
    std::string last_error;
};

namespace {
// single-process RCCL: ncclCommInitAll over the ctx's devices, one all-gather of
// the padded digest slabs.  Resolved with dlopen so that a one-GPU caller never loads RCCL.
struct Rccl {
    void* lib = nullptr;
    bool tried = false, ok = false;
    std::string why;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<ncclComm_t> comms;
};
} // namespace

struct snaphash_batch;

struct snaphash_ctx {
    std::vector<std::unique_ptr<DevCtx>> dev;
    uint32_t host_threads = 0; // host threads the planner may use: 0 = automatic (the cores this process may keep busy)
    bool gpu_only = false;     // SNAPHASH_FLAG_GPU_ONLY: no planner, every byte through the HIP kernels
    unsigned cpus = 1;         // usable_cpus() at init: affinity mask capped by the cgroup's CPU quota
    unsigned cpus_call = 0;    // the cores THIS call may plan with (0 = cpus): a rank of a one-process-per-GPU job plans with its share
    double host_rate = 1.4e9;  // bytes/s of one host thread's SHA-512 on this box, measured at init
    PlanCalib calib;           // the link and the fill threads of THIS box, as measured at init and by every staged call since
    std::mutex calib_mu;       // (the engines of a multi-device ctx report from their own threads)
    uint32_t flags = 0;
    Rccl rccl;
    std::vector<uint8_t*> d_gather; // per device: n_devices * kmax * 64 bytes
    size_t gather_cap = 0;          // rows (kmax) the gather buffers hold
    snaphash_stats stats{};
    snaphash_stats_ex ex{};
    snaphash_targz_stats targz{};
    std::string last_error;
    snaphash_batch* open_batch = nullptr;
    DevCtx* d0() const { return dev[0].get(); }
};

namespace {

int fail(DevCtx* c, int code, const std::string& msg)
{
    if (c) c->last_error = msg;
    return code;
}
int fail(snaphash_ctx* x, int code, const std::string& msg)
{
    if (x) x->last_error = msg;
    return code;
}
// error raised inside an engine -> the ctx's last_error
int lift(snaphash_ctx* x, DevCtx* c, int rc)
{
    if (rc && x && c) x->last_error = c->last_error;
    return rc;
}
#define HIP_TRY(c, expr)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(c, (e_ == hipErrorOutOfMemory) ? SNAPHASH_ENOMEM : SNAPHASH_EDEVICE,         \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                          \
    } while (0)

EventPair* next_events(DevCtx* c, int kind)
{
    if (c->ev_used == c->ev_pool.size()) {
        EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        c->ev_pool.push_back(p);
    }
    EventPair* p = &c->ev_pool[c->ev_used++];
    p->kind = kind;
    return p;
}

void collect_events(DevCtx* c)
{
    static const bool trace = getenv("SNAPHASH_TRACE_EVENTS") != nullptr; // the GPU side of a call, batch by batch
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) == hipSuccess) {
            if (c->ev_pool[i].kind == 0) c->stats.kernel_ms += ms;
            else if (c->ev_pool[i].kind == 1) c->stats.h2d_ms += ms;
        }
        if (trace && c->ev_used > 1) {
            float t0 = 0, t1 = 0;
            (void)hipEventElapsedTime(&t0, c->ev_pool[0].a, c->ev_pool[i].a);
            (void)hipEventElapsedTime(&t1, c->ev_pool[0].a, c->ev_pool[i].b);
            fprintf(stderr, "snaphash engine %d: %s %7.2f .. %7.2f ms\n", c->index, c->ev_pool[i].kind == 0 ? "kernel" : c->ev_pool[i].kind == 1 ? "h2d   " : "other ", t0, t1);
        }
    }
    c->ev_used = 0;
}

int ensure_state(DevCtx* c, size_t n, bool want_digests)
{
    if (n > c->state_cap) {
        if (c->d_state) (void)hipFree(c->d_state);
        c->d_state = nullptr; c->state_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_state, n * 64));
        c->state_cap = n;
    }
    if (want_digests && n > c->digests_cap) {
        if (c->d_digests) (void)hipFree(c->d_digests);
        c->d_digests = nullptr; c->digests_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_digests, n * 64));
        c->digests_cap = n;
    }
    return SNAPHASH_OK;
}

int ensure_jobs(DevCtx* c, Job** h, Job** d, size_t* cap, size_t n)
{
    if (n <= *cap) return SNAPHASH_OK;
    size_t want = std::max<size_t>(n, 1024);
    if (*h) (void)hipHostFree(*h);
    if (*d) (void)hipFree(*d);
    *h = nullptr; *d = nullptr; *cap = 0;
    HIP_TRY(c, hipHostMalloc((void**)h, want * sizeof(Job), hipHostMallocDefault));
    HIP_TRY(c, hipMalloc((void**)d, want * sizeof(Job)));
    *cap = want;
    return SNAPHASH_OK;
}

// Pinned host memory for an engine: on the GPU's NUMA node when it is known (the calling thread's memory policy
// prefers that node for the duration of the allocation and hipHostMallocNumaUser tells the runtime to honour it),
// wherever the runtime puts it otherwise.
hipError_t host_alloc(DevCtx* c, void** p, size_t bytes)
{
    SavedMemPolicy saved;
    if (c->numa_node >= 0 && numa_prefer_node(c->numa_node, &saved)) {
        const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocNumaUser);
        numa_restore_policy(saved);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

// want = 0: the engine's full staging size (every user but the hashing engine, which asks for what its job needs:
// pinning 2 x 256 MiB costs ~40 ms, which a one-shot `snappy build` of a small tree would pay for nothing).
int ensure_slots(DevCtx* c, int nslots = 2, uint64_t want = 0)
{
    if (want == 0 || want > c->staging) want = c->staging;
    for (int k = 0; k < nslots; ++k) {
        Slot& s = c->slot[k];
        if (s.cap < want) {
            if (s.h_buf) (void)hipHostFree(s.h_buf);
            if (s.d_buf) (void)hipFree(s.d_buf);
            s.h_buf = nullptr; s.d_buf = nullptr; s.cap = 0;
            HIP_TRY(c, host_alloc(c, (void**)&s.h_buf, want));
            if (c->staging_node < 0) { s.h_buf[0] = 0; c->staging_node = numa_node_of_address(s.h_buf); }
            HIP_TRY(c, hipMalloc((void**)&s.d_buf, want + 256)); // slack: the deflate kernel peeks 3 bytes past a chunk
            s.cap = want;
        }
        if (!s.done) HIP_TRY(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        if (!s.copied) HIP_TRY(c, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    }
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    return SNAPHASH_OK;
}

// Kernel choice.  PAIR/SPLIT need one workgroup (64 streams, 140 KB of LDS) per CU
// resident at once and pay two block-times of pipeline fill, so they are for few,
// long streams: PAIR holds ~568 GB/s from 16 384 streams on (one workgroup per CU,
// further workgroups queue), WIDE delivers streams x 16 MB/s until it saturates the
// VALUs at 65 536 streams -- the curves cross at ~34 800 streams
// (profiles/r01_regime_sweep.txt); the cap is two full passes of 256 workgroups, because a
// third, mostly empty pass would cost a whole stream-time more.  A heavy-tailed batch (BASELINE config 5: Zipf
// sizes) is cut in two: the long head goes to PAIR (per-stream latency decides the
// makespan), the short tail to WIDE.
constexpr size_t kSplitMaxStreams = 32768;
constexpr uint64_t kSplitMinBlocks = 32;

uint64_t job_blocks(const Job& j) { return (j.nbytes >> 7) + 1; }

hipError_t launch_kernel(uint32_t k, const Job* d_jobs, size_t n, DevCtx* c, uint8_t* d_digests, bool staged)
{
    if (k == SNAPHASH_KERNEL_QUAD) return launch_quad(d_jobs, (uint32_t)n, c->d_state, d_digests, c->stream);
    if (k == SNAPHASH_KERNEL_PAIR) return launch_pair(d_jobs, (uint32_t)n, c->d_state, d_digests, c->stream, staged);
    if (k == SNAPHASH_KERNEL_SPLIT) return launch_split(d_jobs, (uint32_t)n, c->d_state, d_digests, c->stream);
    return launch_wide(d_jobs, (uint32_t)n, c->d_state, d_digests, c->stream);
}

// h_jobs is sorted longest first.  Returns how many leading jobs go to the
// few-long-streams kernel (*k_head) and which kernel takes the rest (*k_tail).
size_t plan_kernels(const DevCtx* c, const Job* h_jobs, size_t n, uint32_t* k_head, uint32_t* k_tail)
{
    *k_tail = SNAPHASH_KERNEL_WIDE;
    if (c->kernel_pref == SNAPHASH_KERNEL_WIDE || c->kernel_pref == SNAPHASH_KERNEL_SPLIT ||
        c->kernel_pref == SNAPHASH_KERNEL_PAIR || c->kernel_pref == SNAPHASH_KERNEL_QUAD) {
        *k_head = c->kernel_pref;
        return n;
    }
    *k_head = SNAPHASH_KERNEL_PAIR;
    uint64_t blocks = 0;
    for (size_t i = 0; i < n; ++i) blocks += job_blocks(h_jobs[i]);
    if (n <= kSplitMaxStreams) {
        if (blocks >= kSplitMinBlocks * n) return n; // few, long streams
        *k_head = SNAPHASH_KERNEL_WIDE;
        return n;
    }
    // many streams: all WIDE unless the head dwarfs the rest
    if (job_blocks(h_jobs[0]) >= 8 * job_blocks(h_jobs[kSplitMaxStreams - 1]) &&
        job_blocks(h_jobs[0]) >= kSplitMinBlocks) {
        size_t head = 0;
        while (head < kSplitMaxStreams && job_blocks(h_jobs[head]) >= kSplitMinBlocks) ++head;
        return head;
    }
    *k_head = SNAPHASH_KERNEL_WIDE;
    return n;
}

// Sort (longest first, so the lanes of a wave finish together), upload and launch.
// When `copied` is given, the job array goes up on the copy stream and the kernel waits
// for that event (the staging engine); otherwise everything is on the launch stream.
int launch_jobs(DevCtx* c, Job* h_jobs, Job* d_jobs, size_t n, uint8_t* d_digests, hipEvent_t copied = nullptr)
{
    if (n == 0) return SNAPHASH_OK;
    std::stable_sort(h_jobs, h_jobs + n, [](const Job& a, const Job& b) { return a.nbytes > b.nbytes; });
    if (copied) {
        HIP_TRY(c, hipMemcpyAsync(d_jobs, h_jobs, n * sizeof(Job), hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(copied, c->copy_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, copied, 0));
    } else {
        HIP_TRY(c, hipMemcpyAsync(d_jobs, h_jobs, n * sizeof(Job), hipMemcpyHostToDevice, c->stream));
    }
    uint32_t k_head, k_tail;
    const size_t head = plan_kernels(c, h_jobs, n, &k_head, &k_tail);
    EventPair* ev = next_events(c, 0);
    if (!ev) return fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
    HIP_TRY(c, hipEventRecord(ev->a, c->stream));
    const bool staged = copied != nullptr; // the job table came through a staging slot
    hipError_t e = launch_kernel(k_head, d_jobs, head, c, d_digests, staged);
    if (e == hipSuccess && head < n) e = launch_kernel(k_tail, d_jobs + head, n - head, c, d_digests, staged);
    if (e != hipSuccess) return fail(c, SNAPHASH_EDEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
    HIP_TRY(c, hipEventRecord(ev->b, c->stream));
    c->stats.launches += (head < n) ? 2 : 1;
    c->stats.kernel_used = k_head;
    c->pending = true;
    return SNAPHASH_OK;
}

int sync_ctx(DevCtx* c)
{
    if (!c->pending && c->ev_used == 0) return SNAPHASH_OK;
    if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    c->pending = false;
    if (c->t_call0 > 0) { c->stats.wall_ms = now_ms() - c->t_call0; c->t_call0 = 0; }
    return SNAPHASH_OK;
}

void begin_call(DevCtx* c)
{
    c->stats = snaphash_stats{};
    c->t_call0 = now_ms();
    c->last_error.clear();
}

// ---- streaming engine: host sources -> staged chunks -> kernels -----------------

struct Source {
    const char* path = nullptr;   // file source
    const uint8_t* mem = nullptr; // memory source
    uint64_t len = 0;
    uint64_t gpu_len = 0;         // bytes this engine hashes: == len (whole stream, digest out) or a multiple of
                                  // 128 below len (prefix only: the chaining value is handed to a host thread)
};

struct ReadOp { uint32_t src; uint64_t off; uint64_t n; uint8_t* dst; bool to_eof; };

// Descriptors a hashing call keeps open between the batches a file appears in: a file of several batches is opened once,
// not once per segment (10 001 x 1 MiB in 64 KiB segments: 160 000 open/close pairs, a third of the fill threads' time).
// A stream has one read operation per batch and batches are filled one after the other, so a slot is touched by one
// thread at a time.  budget: descriptors this engine may hold (RLIMIT_NOFILE, shared among engines); beyond it a
// segment opens and closes its file as before.
struct FdCache {
    std::vector<int> fd;            // per source, -1 = not open
    std::atomic<int64_t> held{0};
    int64_t budget = 0;
    explicit FdCache(size_t n, int64_t b) : fd(n, -1), budget(b) {}
    ~FdCache() { for (int f : fd) if (f >= 0) close(f); }
    FdCache(const FdCache&) = delete;
    FdCache& operator=(const FdCache&) = delete;
};

// One read operation of a staging fill: file source -> open + pread (io.Copy's semantics: a file that shrank or
// grew since its size was taken is an error), memory source -> a streaming copy.  Returns 0 or an errno.
int do_read_op(const Source& s, const ReadOp& op, FdCache* cache = nullptr)
{
    if (s.mem) { copy_to_staging(op.dst, s.mem + op.off, op.n); return 0; }
    int err = 0;
    int fd = cache ? cache->fd[op.src] : -1;
    bool cached = fd >= 0;
    if (fd < 0) {
        // out of descriptors: what holds them (kept descriptors of streams about to end, the host part's lanes) lets go
        // shortly; the reference's one-file-at-a-time loop would not have failed here, so wait a bounded while
        for (unsigned tries = 0;; ++tries) {
            fd = open(s.path, O_RDONLY | O_CLOEXEC);
            if (fd >= 0 || (errno != EMFILE && errno != ENFILE) || tries >= 2000) break;
            usleep(tries < 100 ? 200 : 2000);
        }
        if (fd < 0) err = errno;
        else if (cache && !op.to_eof && cache->held.load(std::memory_order_relaxed) < cache->budget) { // more segments will follow
            cache->fd[op.src] = fd;
            cache->held.fetch_add(1, std::memory_order_relaxed);
            cached = true;
        }
    }
    uint64_t got = 0;
    while (!err && got < op.n) {
        ssize_t r = pread(fd, op.dst + got, op.n - got, (off_t)(op.off + got));
        if (r < 0) { if (errno == EINTR) continue; err = errno; }
        else if (r == 0) err = EIO; // file shrank underneath us
        else got += (uint64_t)r;
    }
    if (!err && op.to_eof) { // io.Copy reads to EOF: a file that grew since its size was taken is an error too
        uint8_t probe;
        ssize_t r;
        do r = pread(fd, &probe, 1, (off_t)(op.off + op.n)); while (r < 0 && errno == EINTR);
        if (r > 0) err = EIO;
        else if (r < 0) err = errno;
    }
    if (fd >= 0 && (!cached || op.to_eof || err)) { // the stream's last segment (or an error) closes a kept descriptor
        close(fd);
        if (cached) { cache->fd[op.src] = -1; cache->held.fetch_sub(1, std::memory_order_relaxed); }
    }
    return err;
}

// Fills a staging slot: the engine's pool of fill threads (on the GPU's NUMA node, hostfill.h) runs the operations.
void run_reads(DevCtx* c, const std::vector<Source>& src, const std::vector<ReadOp>& ops, std::atomic<int>& first_err,
               std::atomic<int64_t>& first_err_src, FdCache* cache = nullptr, bool alone = false)
{
    // Staging-fill threads, measured on the GPU box (tools/copy_threads_sweep.sh): copies from caller memory
    // peak at 6 threads (44 GiB/s end to end; 16 threads: 33 -- they fight the concurrent H2D DMA for host
    // memory bandwidth), pread of files at 12 (30 GiB/s).  SNAPHASH_COPY_THREADS overrides (1..256); a ctx with
    // several engines divides the CPUs it may use among them (fill_cap).
    static const int forced = [] {
        const char* e = getenv("SNAPHASH_COPY_THREADS");
        const int v = e ? atoi(e) : 0;
        return (v >= 1 && v <= 256) ? v : 0;
    }();
    const bool from_memory = !ops.empty() && src[ops[0].src].mem != nullptr;
    unsigned cap = forced ? (unsigned)forced : std::min(c->fill_cap, from_memory ? 6u : 12u);
    if (!forced && c->fill_call_cap) cap = std::min(cap, c->fill_call_cap);
    // the first fill of a call has the machine to itself (no copy in flight to fight for memory bandwidth, no kernel to feed):
    // every core the engine may use takes part
    if (!forced && alone && !c->fill_call_cap && c->owner) cap = std::max(cap, std::min(c->fill_cap * 2u, c->owner_cpus_per_engine));
    const unsigned T = (unsigned)std::min<size_t>(cap, std::max<size_t>(1, ops.size() / 4));
    const double t_fill0 = now_ms();
    struct FillAccount { // what the fill threads of this box move (planner.h PlanCalib): wall x threads, bytes
        DevCtx* c; const std::vector<ReadOp>& ops; unsigned T; double t0;
        ~FillAccount()
        {
            uint64_t b = 0;
            for (const ReadOp& op : ops) b += op.n;
            c->fill_bytes += b;
            c->fill_thread_s += (now_ms() - t0) * 1e-3 * T;
        }
    } account{c, ops, T, t_fill0};
    c->pool.parallel_for(ops.size(), T, [&](size_t i) {
        if (first_err.load(std::memory_order_relaxed)) return;
        const ReadOp& op = ops[i];
        const int err = do_read_op(src[op.src], op, cache);
        if (err) {
            int z = 0;
            if (first_err.compare_exchange_strong(z, err)) first_err_src.store(op.src);
        }
    });
}

// Hashes src[i].gpu_len bytes of every source on this engine's device.  digests (host, n*64, may be NULL):
// receives the digests of whole streams; with NULL they stay in c->d_digests (row i = source i) for the
// multi-device gather.  states (host, n*8 u64, may be NULL): receives the chaining values, needed for
// prefix-only sources.  err_src: index of the failing source.
int hash_sources(DevCtx* c, const std::vector<Source>& src, uint8_t* digests, uint64_t* states, int* err_no,
                 int64_t* err_src)
{
    const size_t n = src.size();
    if (err_no) *err_no = 0;
    if (err_src) *err_src = -1;
    if (n == 0) return SNAPHASH_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    uint64_t job_bytes = 0;
    for (const Source& sc : src) job_bytes += sc.gpu_len;
    const unsigned nslots = job_bytes > 2 * c->staging ? 3u : 2u;
    // slots as large as the job needs, in powers of two from 8 MiB up to the engine's staging size (they grow when a
    // larger job comes by, and never shrink)
    uint64_t slot_bytes = std::min<uint64_t>(c->staging, 8u << 20);
    while (slot_bytes < c->staging && slot_bytes < job_bytes / 2 + kAlign * n) slot_bytes <<= 1;
    slot_bytes = std::min(slot_bytes, c->staging);
    for (unsigned k = 0; k < nslots; ++k) slot_bytes = std::max(slot_bytes, std::min(c->slot[k].cap, c->staging)); // what is there already is used
    int rc = ensure_slots(c, (int)nslots, slot_bytes);
    if (rc) return rc;
    rc = ensure_state(c, n, true);
    if (rc) return rc;

    c->fill_thread_s = 0;
    c->fill_bytes = 0;
    std::vector<uint64_t> done(n, 0);
    std::vector<uint32_t> active;
    active.reserve(n);
    for (size_t i = 0; i < n; ++i)
        if (src[i].gpu_len > 0 || src[i].len == 0) active.push_back((uint32_t)i); // a prefix of 0 bytes needs no launch
    // longest first: the streams that set the makespan are served in every batch from the first one on
    std::stable_sort(active.begin(), active.end(), [&](uint32_t a, uint32_t b) { return src[a].gpu_len > src[b].gpu_len; });
    std::atomic<int> first_err{0};
    std::atomic<int64_t> first_err_src{-1};
    std::vector<ReadOp> ops;
    const bool from_memory = src[0].mem != nullptr;
    FdCache fds(from_memory ? 0 : n, c->fd_call_budget >= 0 ? std::min(c->fd_call_budget, c->fd_budget) : c->fd_budget);
    const uint64_t seg_floor = from_memory ? kMinSegmentMem : kMinSegment;
    // The batch: a job of more than a buffer is cut into about two dozen batches (32 MiB at least, a buffer at most, and
    // room for every stream's floor), each in a sub-slot of the buffers; so many are in flight that the fill runs
    // ahead of the copy engine and the copy engine ahead of the kernels.
    uint64_t S_full = slot_bytes;
    if (job_bytes + kAlign * n > slot_bytes && (slot_bytes & (slot_bytes - 1)) == 0 && slot_bytes > (32u << 20)) {
        S_full = 32u << 20;
        while (S_full < slot_bytes && (S_full < job_bytes / 24 || S_full < (seg_floor + kAlign) * std::min<uint64_t>(n, kTargetStreams))) S_full <<= 1;
    }
    const unsigned per_slot = (unsigned)(slot_bytes / S_full), nsub = nslots * per_slot;
    if (c->sub.size() < nsub) c->sub.resize(nsub);
    for (unsigned q = 0; q < nsub; ++q) {
        SubSlot& ss = c->sub[q];
        if (!ss.done) HIP_TRY(c, hipEventCreateWithFlags(&ss.done, hipEventDisableTiming));
        if (!ss.copied) HIP_TRY(c, hipEventCreateWithFlags(&ss.copied, hipEventDisableTiming));
        ss.busy = false;
    }
    const size_t n_active0 = active.size();
    // lab knobs, read once a CALL (so that one process can alternate settings between passes: tools/ramp_ab.py)
    const size_t new_cap = [] { // streams a batch may BEGIN (file sources, trees of more than 2 048 streams); SNAPHASH_NEW_PER_BATCH=0: no cap
        const char* e = getenv("SNAPHASH_NEW_PER_BATCH");
        return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)1024;
    }();
    const bool hold_back_on = [] { const char* e = getenv("SNAPHASH_HOLD_BACK"); return !e || atoi(e) != 0; }();
    bool held_back = false; // the last batch of a link-bound job has been cut in two (below)
    const unsigned ramp_shift = [] { // the first batch of a large job is S_full >> this
        const char* e = getenv("SNAPHASH_RAMP_SHIFT");
        const unsigned long v = e ? strtoul(e, nullptr, 10) : 3ul;
        return (unsigned)std::min<unsigned long>(std::max<unsigned long>(v, 1), 8);
    }();
    // ... of a job of many streams (link-bound), "first fraction in 1/64ths, growth per batch in percent": 24,115 = three
    // eighths of a batch first, 15 % more each time (below)
    unsigned ramp_first64 = 24, ramp_growth_pct = 115;
    if (const char* e = getenv("SNAPHASH_RAMP_MANY")) {
        unsigned a = 0, b = 0;
        if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 1 && a <= 64 && b >= 100 && b <= 400) { ramp_first64 = a; ramp_growth_pct = b; }
    }
    unsigned batch = 0;
    double t_wait = 0, t_plan = 0, t_read = 0, t_launch = 0; // where the host side of the engine spends its time (SNAPHASH_TRACE_TREE)
    const double t_engine0 = now_ms();
    double t_first_copy = 0; // host clock: the first H2D is enqueued this long after the engine started
    size_t n_copies = 0;     // H2D copies of this call (what its link observation is worth: planner.h PlanCalib::observe_call)

    while (!active.empty()) {
        const unsigned q = batch % nsub;
        SubSlot& sl = c->sub[q];
        uint8_t* const sl_h = c->slot[q / per_slot].h_buf + (uint64_t)(q % per_slot) * S_full;
        uint8_t* const sl_d = c->slot[q / per_slot].d_buf + (uint64_t)(q % per_slot) * S_full;
        const double tb0 = now_ms();
        if (sl.busy) { HIP_TRY(c, hipEventSynchronize(sl.done)); sl.busy = false; }
        const double tb1 = now_ms();
        t_wait += tb1 - tb0;
        // What a stream gets of this slot: its share by remaining length (so that long and short streams end in the
        // same batch -- a long stream served a fixed slice per batch would still be running, alone, long after the
        // others: the per-stream rate of the kernels is what it is), but at least a floor (a file is opened once per
        // batch it appears in).  Equal streams (config 2) fill a slot kTargetStreams at a time, as before.
        // Both ends of a job of several slots are tapered: nothing overlaps the first fill and the
        // first copy, and nothing overlaps the last copy and the last kernel, so the first batches are 1/8, 1/4, 1/2 of
        // a slot and the last ones halve what is left -- invisible on a 10 GiB job, a fifth of the time of the 1.3 GiB
        // shard one of eight ranks gets.  A job that fits one slot is one batch.
        long double total_rem = 0;
        size_t n_started = 0;
        for (uint32_t id : active) { total_rem += (long double)(src[id].gpu_len - done[id]); n_started += done[id] != 0; }
        uint64_t S = S_full;
        if (job_bytes + kAlign * n > S_full) {
            if (n_active0 > 2048 && ramp_growth_pct > 100) {
                // Many streams: the link is the bound, and the fill threads are only ~1.3 x as fast as the link (75 against
                // 57 GB/s).  The link idles while the first batch is filled, and again before every batch that takes longer
                // to fill than its predecessor takes to copy -- doubling batches (rounds 1-4: 1/8, 1/4, 1/2, 1) lose
                // S x (2 / 75 - 1 / 57 GB/s) at every step: 3.5 ms in all on config 2, measured 3.6
                // (profiles/r05_tree_events_before.txt).  A batch that grows by less than fill rate / link rate a step never
                // makes the link wait: three eighths of a batch first, 15 % more each time (tools/ramp_ab.py; profiles/r05_ramp.txt:
                // the link idle 1.4 + 1.1 ms at the start instead of 1.0 + 4.8).
                double f = (double)ramp_first64 / 64.0;
                for (unsigned k = 0; k < batch && f < 1.0; ++k) f *= (double)ramp_growth_pct / 100.0;
                if (f < 1.0) S = std::max<uint64_t>((uint64_t)((double)S_full * f) & ~(uint64_t)(kAlign - 1), std::min<uint64_t>(S_full, 1u << 20));
            } else if (batch < ramp_shift) {
                // ... but never so small that only some streams get their floor: a batch costs the kernel chain its LARGEST
                // share's time, so 256 streams at 32 KiB cost what all 1 250 at 32 KiB would (the file-source shard's first
                // three batches: 0.75 ms of kernel each for 8, 16 and 32 MiB; profiles/r04_shard_trace.txt)
                // (That is a concern of jobs bound by their kernel chain: up to ~2 000 streams, whose 44 MB/s each do not
                // outrun the link.  With more streams the link is the bound and the first copy should start early: the C2
                // tree's first batch was a whole 256 MiB buffer, 4 ms of fill with the link idle.)
                // (Finer steps -- x 1.4 a batch from 16 MiB, eight of them -- were tried for that regime and left the link idle
                // MORE, 4.5 ms against 3.5: every batch costs ~0.4 ms of planning and hand-over whatever its size.)
                const uint64_t every = active.size() <= 2048 ? std::min<uint64_t>(S_full, (seg_floor + kAlign) * (uint64_t)active.size()) : 0;
                S = std::max<uint64_t>({(S_full >> (ramp_shift - batch)) & ~(uint64_t)(kAlign - 1), every & ~(uint64_t)(kAlign - 1), std::min<uint64_t>(S_full, 1u << 20)});
            }
            if (total_rem < 2 * (long double)S) { // the end: half of what is left, while every stream can still get its floor
                const uint64_t half = ((uint64_t)(total_rem / 2) + kAlign * active.size()) & ~(uint64_t)(kAlign - 1);
                const uint64_t least = std::max<uint64_t>(S_full >> 5, (seg_floor + kAlign) * active.size());
                if (half >= least) S = std::min(S, half);
            }
        }
        // A file's FIRST segment costs an open(), and every open of a process takes the lock of its one descriptor table
        // (~3 us alone, ~19 us each with twelve threads at it).  A tree of many files used to begin all of
        // them within its first four batches -- config 2: 10 001 opens in the first 480 MiB, whose fills ran at 40-50 GB/s
        // where later ones run at 76, and the link idled 6 ms of the ramp (profiles/r05_tree_events_before.txt).  So a
        // batch begins at most new_cap streams; the rest of it goes to streams already open (kept descriptors: a pread
        // each).  Streams nobody has begun go in front of those served at the floor, so every batch begins its share.
        const bool cap_new = new_cap != 0 && !from_memory && n_active0 > 2048;
        const size_t n_serve = cap_new ? std::min<size_t>(kTargetStreams, n_started + std::min<size_t>(new_cap, active.size() - n_started)) : kTargetStreams;
        const uint64_t floor_q = std::max<uint64_t>(seg_floor, (S / std::max<size_t>(1, n_serve)) & ~(uint64_t)(kAlign - 1));
        // What the shares are taken of: the batch less the alignment every segment may cost.  Without that the shares of
        // ALL streams came to a whole batch, the padding pushed the last dozen streams of the list out of every batch, and
        // they were hashed at the end, alone, at 44 MB/s each (5 000 x 1 MiB: the last four kernels took 26 ms instead of
        // 6, 114 ms for a job whose copies take 92; profiles/r04_shard_trace.txt).
        const uint64_t pad = kAlign * (uint64_t)active.size();
        const uint64_t S_share = pad < S / 2 ? S - pad : S;
        rc = ensure_jobs(c, &sl.h_jobs, &sl.d_jobs, &sl.jobs_cap, active.size());
        if (rc) return rc;

        ops.clear();
        size_t nj = 0;
        uint64_t used = 0;
        // Who is served next time.  A batch cannot always serve every stream (more streams than it has floors for: 5 000 x
        // 1 MiB at a floor of 64 KiB, 100 000 small files); round 3 then served the SAME leading streams batch after batch and
        // the ones behind them only when those were done -- 904 of 5 000 streams hashed at the end, alone, 290 KiB a batch
        // at 44 MB/s: kernels of 6-9 ms behind copies of 4.7 (profiles/r04_shard_trace.txt).  Now the streams a batch had
        // no room for go FIRST in the next one, in front of those that were served at the floor; streams whose share by
        // length exceeds the floor (the long ones that set the makespan) stay in front of both and are served every time.
        std::vector<uint32_t> still, skipped, floor_still;
        still.reserve(active.size());
        // The last batch of a link-bound job: nothing overlaps its kernel, which takes what its LARGEST share takes at a
        // stream's 44 MB/s -- 64 KiB shares: 1.5-1.7 ms behind the last copy (profiles/r05_tree_events_before.txt).  So the batch
        // that would be the last leaves 16 KiB of every stream behind for one more, whose kernel is 0.4 ms.
        constexpr uint64_t kHold = 16u << 10;
        // (... of streams that HAVE that much left -- more than the 24 KiB below which a stream goes to the batch behind whole, on
        // average: 5 000 x 8 KiB made the batch in front of the last an EMPTY one, profiles/r05_small_files.txt.  Config 2's last
        // batch has 26 KiB a stream: a first version of this test asked for 32 and switched the hold-back off for it, +1.5 ms.)
        const bool hold_back = hold_back_on && !held_back && n_active0 > 2048 && total_rem <= (long double)S_share && total_rem > (long double)(8u << 20) &&
                               total_rem > (long double)(kHold + kHold / 2) * (long double)active.size();
        if (hold_back) held_back = true;
        bool full = false;
        size_t n_new = 0;
        // (one division a batch, not one a stream: the engine's thread plans 4 096 segments a batch between two fills, and in
        // the ramp nothing hides that)
        const bool share_all = total_rem > (long double)S;
        const long double share_ratio = share_all ? (long double)S_share / total_rem : 1.0L;
        for (size_t ai = 0; ai < active.size(); ++ai) {
            const uint32_t id = active[ai];
            if (full) { skipped.insert(skipped.end(), active.begin() + (ptrdiff_t)ai, active.end()); break; } // nobody behind a full batch is looked at
            if (cap_new && done[id] == 0 && src[id].gpu_len != 0) {
                if (n_new >= new_cap) { skipped.push_back(id); continue; } // begun by a later batch
                ++n_new;
            }
            const uint64_t rem = src[id].gpu_len - done[id];
            uint64_t quota = share_all ? (uint64_t)((long double)rem * share_ratio) : rem;
            const bool at_floor = (quota & ~(uint64_t)(kAlign - 1)) < floor_q;
            quota = std::max(quota & ~(uint64_t)(kAlign - 1), floor_q); // a multiple of 128: segments are whole blocks
            uint64_t take = rem <= quota ? rem : quota;
            if (hold_back) {
                if (rem <= kHold + kHold / 2) { skipped.push_back(id); continue; } // all of it in the batch behind this one
                take = std::min<uint64_t>(take, (rem - kHold) & ~(uint64_t)(kAlign - 1));
            }
            const uint64_t at = (used + kAlign - 1) & ~(uint64_t)(kAlign - 1);
            if (at + take > S) { full = true; skipped.push_back(id); continue; }
            const bool last = take == rem;
            const bool fin = last && src[id].gpu_len == src[id].len;
            Job j;
            j.data = (uint64_t)(uintptr_t)(sl_d + at);
            j.nbytes = take;
            j.total_prev = done[id];
            j.idx = id;
            j.flags = (done[id] == 0 ? kJobFirst : 0u) | (fin ? kJobFinal : 0u);
            sl.h_jobs[nj++] = j;
            if (take) ops.push_back(ReadOp{id, done[id], take, sl_h + at, fin && src[id].path != nullptr});
            used = at + take;
            done[id] += take;
            c->stats.blocks += padded_blocks(take, fin);
            if (!last) (at_floor ? floor_still : still).push_back(id);
        }
        still.insert(still.end(), skipped.begin(), skipped.end());
        still.insert(still.end(), floor_still.begin(), floor_still.end());
        active.swap(still);
        static const bool trace_batches = getenv("SNAPHASH_TRACE_BATCHES") != nullptr; // (read once, not once a batch)
        if (trace_batches) {
            uint64_t mx = 0;
            for (size_t k = 0; k < nj; ++k) mx = std::max<uint64_t>(mx, sl.h_jobs[k].nbytes);
            fprintf(stderr, "snaphash engine %d: batch %u: S %llu, %zu segments, %llu bytes, largest share %llu, %zu streams left behind\n", c->index, batch,
                    (unsigned long long)S, nj, (unsigned long long)used, (unsigned long long)mx, active.size());
        }
        const double tb2 = now_ms();
        t_plan += tb2 - tb1;

        run_reads(c, src, ops, first_err, first_err_src, from_memory ? nullptr : &fds, batch == 0);
        if (first_err.load()) break;
        const double tb3 = now_ms();
        t_read += tb3 - tb2;
        if (trace_batches)
            fprintf(stderr, "snaphash engine %d: batch %u: at %.2f ms: waited %.2f, planned %.2f, filled %.2f ms (%zu streams begun, %.1f GB/s)\n", c->index, batch,
                    tb0 - t_engine0, tb1 - tb0, tb2 - tb1, tb3 - tb2, n_new, (double)used / ((tb3 - tb2) * 1e6 + 1e-9));

        if (used) { // copy stream: the slot's previous kernel was already waited for above
            if (batch == 0) t_first_copy = now_ms() - t_engine0;
            ++n_copies;
            EventPair* ev = next_events(c, 1);
            if (!ev) return fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
            HIP_TRY(c, hipEventRecord(ev->a, c->copy_stream));
            HIP_TRY(c, hipMemcpyAsync(sl_d, sl_h, used, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(c, hipEventRecord(ev->b, c->copy_stream));
        }
        rc = launch_jobs(c, sl.h_jobs, sl.d_jobs, nj, c->d_digests, sl.copied);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(sl.done, c->stream));
        sl.busy = true;
        ++batch;
        t_launch += now_ms() - tb3;
    }

    const double ts0 = now_ms();
    static const bool trace_tree = getenv("SNAPHASH_TRACE_TREE") != nullptr;
    float gpu_span = 0, copy_busy = 0, copy_span = 0; // first copy's start -> last kernel's end; the copies' own time; first copy's start -> last copy's end
    if (trace_tree && c->ev_used > 1) {
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->stream);
        size_t first_copy = c->ev_used, last_copy = 0, last_kernel = 0;
        for (size_t i = 0; i < c->ev_used; ++i) {
            float ms = 0;
            if (c->ev_pool[i].kind == 1) {
                if (first_copy == c->ev_used) first_copy = i;
                last_copy = i;
                if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) == hipSuccess) copy_busy += ms;
            } else if (c->ev_pool[i].kind == 0) last_kernel = i;
        }
        if (first_copy < c->ev_used) {
            (void)hipEventElapsedTime(&gpu_span, c->ev_pool[first_copy].a, c->ev_pool[last_kernel].b);
            (void)hipEventElapsedTime(&copy_span, c->ev_pool[first_copy].a, c->ev_pool[last_copy].b);
        }
    }
    rc = sync_ctx(c);
    if (trace_tree)
        fprintf(stderr, "snaphash engine %d: %u batches; waiting for a slot %.1f ms, planning %.1f ms, reads %.1f ms, enqueue %.1f ms, drain %.1f ms; "
                        "first copy enqueued at %.2f ms, copies busy %.2f of %.2f ms, first copy -> last kernel %.2f ms, engine %.2f ms\n",
                c->index, batch, t_wait, t_plan, t_read, t_launch, now_ms() - ts0, t_first_copy, copy_busy, copy_span, gpu_span, now_ms() - t_engine0);
    for (SubSlot& ss : c->sub) ss.busy = false;
    if (rc) return rc;
    if (first_err.load()) {
        const int64_t s = first_err_src.load();
        if (err_no) *err_no = first_err.load();
        if (err_src) *err_src = s;
        return fail(c, SNAPHASH_EIO,
                    std::string(s >= 0 && src[s].path ? src[s].path : "<buffer>") + ": " + strerror(first_err.load()));
    }
    if (digests) HIP_TRY(c, hipMemcpy(digests, c->d_digests, n * 64, hipMemcpyDeviceToHost));
    if (states) HIP_TRY(c, hipMemcpy(states, c->d_state, n * 64, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) c->stats.bytes_hashed += src[i].gpu_len;
    c->stats.streams = n;
    if (c->owner) { // what this call says about the box: the copies' own rate (HIP events) and what a fill thread moved
        std::lock_guard<std::mutex> lk(c->owner->calib_mu);
        // (the fill: not the engine's first staged call -- its fill threads are being created, its staging pages touched for the
        // first time -- nor a call that took three times what its plan said: whatever happened there (cold page cache, a
        // neighbour on the box) is not what the next call will meet)
        const double planned = c->owner->ex.planned_gpu_ms, took = now_ms() - t_engine0;
        const bool take_fill = c->staged_calls++ > 0 && !(planned > 0 && took > 3.0 * planned);
        c->owner->calib.observe_call(!from_memory, (double)c->fill_bytes, (double)n, (double)n_copies, c->stats.h2d_ms * 1e-3, c->fill_thread_s, take_fill);
    }
    return SNAPHASH_OK;
}

// ---- RCCL gather of the digest vector (several devices) ---------------------------------

bool rccl_load(snaphash_ctx* x)
{
    Rccl& r = x->rccl;
    if (r.tried) return r.ok;
    r.tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.why = "librccl.so not found"; return false; }
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
        r.why = "librccl.so lacks a required symbol";
        return false;
    }
    std::vector<int> devs;
    for (auto& d : x->dev) devs.push_back(d->device);
    std::vector<int> uniq = devs;
    std::sort(uniq.begin(), uniq.end());
    if (std::adjacent_find(uniq.begin(), uniq.end()) != uniq.end()) {
        r.why = "a device ordinal repeats in the ctx (RCCL needs one rank per GPU)";
        return false;
    }
    r.comms.assign(devs.size(), nullptr);
    const ncclResult_t e = r.CommInitAll(r.comms.data(), (int)devs.size(), devs.data());
    if (e != ncclSuccess) {
        r.why = std::string("ncclCommInitAll: ") + (r.GetErrorString ? r.GetErrorString(e) : "error");
        r.comms.clear();
        return false;
    }
    r.ok = true;
    return true;
}

// Every device d holds cnt[d] digests in dev[d]->d_digests (rows 0..cnt[d]-1).  Gathers them into `rows`
// (host, ndev * kmax * 64): device d's rows start at d * kmax.
int gather_digest_slabs(snaphash_ctx* x, const std::vector<size_t>& cnt, size_t kmax, std::vector<uint8_t>& rows)
{
    const size_t nd = x->dev.size();
    rows.assign(nd * kmax * 64, 0);
    const double t0 = now_ms();
    bool use_rccl = !(x->flags & SNAPHASH_FLAG_NO_RCCL) && rccl_load(x);
    if (use_rccl) {
        if (kmax > x->gather_cap) {
            for (size_t d = 0; d < x->d_gather.size(); ++d)
                if (x->d_gather[d]) { (void)hipSetDevice(x->dev[d]->device); (void)hipFree(x->d_gather[d]); }
            x->d_gather.assign(nd, nullptr);
            x->gather_cap = 0;
            for (size_t d = 0; d < nd; ++d) {
                DevCtx* c = x->dev[d].get();
                HIP_TRY(c, hipSetDevice(c->device));
                if (hipMalloc((void**)&x->d_gather[d], nd * kmax * 64) != hipSuccess)
                    return fail(x, SNAPHASH_ENOMEM, "hipMalloc of the gather buffer failed");
            }
            x->gather_cap = kmax;
        }
        for (size_t d = 0; d < nd; ++d) { // already sized by the caller before hashing (a growth here would drop the digests)
            DevCtx* c = x->dev[d].get();
            HIP_TRY(c, hipSetDevice(c->device));
            int rc = ensure_state(c, kmax, true);
            if (rc) return lift(x, c, rc);
        }
        Rccl& r = x->rccl;
        ncclResult_t e = r.GroupStart();
        for (size_t d = 0; d < nd && e == ncclSuccess; ++d) {
            DevCtx* c = x->dev[d].get();
            (void)hipSetDevice(c->device);
            e = r.AllGather(c->d_digests, x->d_gather[d], kmax * 64, ncclUint8, r.comms[d], c->stream);
        }
        const ncclResult_t e2 = r.GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) {
            // The digests are all there, each on its device: a collective that will not run costs the collective, not
            // the pass.  Remember why (snaphash_stats_ex.gather_kind says which way the vector came) and copy.
            r.ok = false;
            r.why = std::string("RCCL all-gather: ") + (r.GetErrorString ? r.GetErrorString(e) : "error");
            use_rccl = false;
        }
    }
    if (use_rccl) {
        Rccl& r = x->rccl;
        (void)r;
        for (size_t d = 0; d < nd; ++d) {
            DevCtx* c = x->dev[d].get();
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        DevCtx* c0 = x->d0();
        HIP_TRY(c0, hipSetDevice(c0->device));
        HIP_TRY(c0, hipMemcpy(rows.data(), x->d_gather[0], nd * kmax * 64, hipMemcpyDeviceToHost));
        x->ex.gather_kind = 1;
        if (x->flags & SNAPHASH_FLAG_CHECK_GATHER) { // the collective's parity check: per-device copies
            std::vector<uint8_t> own(kmax * 64);
            for (size_t d = 0; d < nd; ++d) {
                DevCtx* c = x->dev[d].get();
                HIP_TRY(c, hipSetDevice(c->device));
                if (cnt[d]) HIP_TRY(c, hipMemcpy(own.data(), c->d_digests, cnt[d] * 64, hipMemcpyDeviceToHost));
                if (memcmp(own.data(), rows.data() + d * kmax * 64, cnt[d] * 64) != 0)
                    return fail(x, SNAPHASH_EDEVICE, "RCCL-gathered digest slab differs from the device's own copy");
            }
            x->ex.gather_checked = 1;
        }
    } else {
        for (size_t d = 0; d < nd; ++d) {
            DevCtx* c = x->dev[d].get();
            HIP_TRY(c, hipSetDevice(c->device));
            if (cnt[d]) HIP_TRY(c, hipMemcpy(rows.data() + d * kmax * 64, c->d_digests, cnt[d] * 64, hipMemcpyDeviceToHost));
        }
        x->ex.gather_kind = 2;
    }
    x->ex.gather_ms = now_ms() - t0;
    return SNAPHASH_OK;
}

// ---- planning (planner.h): which streams the kernels take and which the library's own host SHA-512 ------

// bytes per second of the library's host SHA-512 on one core of THIS box (hosts differ 3x): timed on a 512 KiB buffer when a
// ctx that may plan is created (~2 ms), the MEDIAN of five -- and smoothed over the ctxs of the process (the first stands, later
// ones move it a quarter of the way): one bench run's ctxs measured 1.22 .. 1.94 GB/s on one box (a core still clocking up,
// a sibling thread busy, the buffer hot in L2), config 5's 255 MiB head was planned at 218 ms where it takes 190, and a
// 3 MiB file at 1.6 ms where it takes 2.4.  observe_host corrects the model only over calls of 5 ms and more.
double measure_host_rate()
{
    static std::mutex mu;
    static double smoothed = 0; // process-wide
    std::vector<uint8_t> buf(512u << 10, 0x5a);
    HostSha hs;
    uint8_t out[64];
    double rate[5];
    int n = 0;
    for (int rep = 0; rep < 5; ++rep) {
        host_sha512_init(hs);
        const double t0 = now_ms();
        host_sha512_update(hs, buf.data(), buf.size());
        host_sha512_final(hs, out);
        const double dt = (now_ms() - t0) * 1e-3;
        if (dt > 0) rate[n++] = (double)buf.size() / dt;
    }
    if (n == 0) return 0.40e9;
    std::sort(rate, rate + n);
    const double median = rate[n / 2];
    if (!(median > 50e6)) return 0.40e9;
    std::lock_guard<std::mutex> lk(mu);
    smoothed = smoothed > 0 ? 0.75 * smoothed + 0.25 * median : median;
    return smoothed;
}

// descriptors this process holds right now (the entries of /proc/self/fd, less the one the listing itself uses); 0 = unknown
int64_t open_descriptors()
{
    DIR* d = opendir("/proc/self/fd");
    if (!d) return 0;
    int64_t n = 0;
    while (struct dirent* e = readdir(d))
        if (e->d_name[0] != '.') ++n;
    closedir(d);
    return n > 0 ? n - 1 : 0;
}

PlanModel plan_model_of(const snaphash_ctx* x, bool from_files)
{
    PlanModel m;
    m.n_devices = (unsigned)x->dev.size();
    m.cpus = x->cpus_call ? x->cpus_call : x->cpus;
    m.fill_threads = std::min(x->d0()->fill_cap, from_files ? 12u : 6u); // what run_reads uses per engine
    if (x->cpus_call) m.fill_threads = std::max(1u, std::min(m.fill_threads, x->cpus_call > 2u ? x->cpus_call - 1u : 1u));
    if (host_sha512_x8_available()) m.host_lane_gain = from_files ? 2.4 : 3.2; // (measured on the box per pool thread: 3.1 GB/s of files, 4.5 GB/s of memory against 1.26 / 1.4 one stream at a time)
    m.host_threads = x->host_threads;
    m.from_files = from_files;
    m.host_rate = from_files ? x->host_rate * 0.9 : x->host_rate; // a host thread reads its file itself (pread, then hash)
    {
        std::lock_guard<std::mutex> lk(const_cast<snaphash_ctx*>(x)->calib_mu);
        x->calib.apply(m); // this box's link and fill rates where they have been measured (planner.h PlanCalib)
    }
    return m;
}

// ---- one call: hybrid split, LPT shards, per-device engines, gather ------------------------

void merge_stats(snaphash_ctx* x)
{
    snaphash_stats t{};
    for (auto& d : x->dev) {
        const snaphash_stats& s = d->stats;
        t.bytes_hashed += s.bytes_hashed;
        t.blocks += s.blocks;
        t.streams += s.streams;
        t.launches += s.launches;
        if (s.launches) t.kernel_used = s.kernel_used;
        t.kernel_ms = std::max(t.kernel_ms, s.kernel_ms);
        t.h2d_ms = std::max(t.h2d_ms, s.h2d_ms);
    }
    x->ex.gpu_bytes = t.bytes_hashed;
    t.bytes_hashed += x->ex.host_bytes;
    t.streams += x->ex.host_streams;
    t.wall_ms = x->stats.wall_ms;
    x->stats = t;
}

void begin_top(snaphash_ctx* x)
{
    x->stats = snaphash_stats{};
    x->ex = snaphash_stats_ex{};
    x->ex.struct_size = sizeof(snaphash_stats_ex);
    x->ex.n_devices = (uint32_t)x->dev.size();
    x->last_error.clear();
    for (auto& d : x->dev) { begin_call(d.get()); d->t_call0 = 0; }
}

bool keep_on_device_planned(const snaphash_ctx* x, size_t nd) { return nd > 1 || (x->flags & SNAPHASH_FLAG_FORCE_GATHER); }

int hash_sources_top(snaphash_ctx* x, std::vector<Source>& src, uint8_t* digests, int32_t* status)
{
    const size_t n = src.size();
    if (status) for (size_t i = 0; i < n; ++i) status[i] = 0;
    if (n == 0) return SNAPHASH_OK;
    const size_t nd = x->dev.size();
    for (Source& s : src) s.gpu_len = s.len;
    std::vector<uint8_t> on_host(n, 0);
    unsigned plan_threads = 0;
    const double t_hash0 = now_ms();
    struct HashClock { // snaphash_stats_ex.hash_ms on every way out
        snaphash_ctx* x; double t0;
        ~HashClock() { x->ex.hash_ms = now_ms() - t0; }
    } hash_clock{x, t_hash0};
    if (!x->gpu_only) {
        std::vector<uint64_t> lens(n);
        for (size_t i = 0; i < n; ++i) lens[i] = src[i].len;
        PlanResult plan = plan_streams(lens.data(), n, plan_model_of(x, src[0].path != nullptr));
        on_host.swap(plan.on_host);
        plan_threads = plan.host_threads;
        // the prediction, to be read beside what the call then takes (gpu_ms, host_ms below)
        x->ex.planned_gpu_ms = plan.gpu_seconds * 1e3;
        x->ex.planned_host_ms = plan.host_seconds * 1e3;
        x->ex.planned_threads = plan.host_threads;
        x->ex.plan_ms = now_ms() - t_hash0;
    }

    // GPU part: LPT over the devices by SHA-512 block count (deterministic)
    std::vector<uint32_t> gidx;
    std::vector<uint32_t> hidx;
    for (size_t i = 0; i < n; ++i) (on_host[i] ? hidx : gidx).push_back((uint32_t)i);
    std::vector<std::vector<uint32_t>> member(nd);
    if (nd == 1) {
        member[0] = gidx;
    } else {
        std::vector<uint64_t> lens(gidx.size());
        for (size_t k = 0; k < gidx.size(); ++k) lens[k] = src[gidx[k]].len;
        std::vector<int32_t> shard(gidx.size());
        if (!gidx.empty()) lpt_assign(lens.data(), lens.size(), (int)nd, shard.data());
        for (size_t k = 0; k < gidx.size(); ++k) member[shard[k]].push_back(gidx[k]);
    }

    if (keep_on_device_planned(x, nd)) { // the slab every device contributes to the gather is kmax rows: size it BEFORE hashing
        size_t kmax = 1;
        for (size_t d = 0; d < nd; ++d) kmax = std::max(kmax, member[d].size());
        for (size_t d = 0; d < nd; ++d) {
            DevCtx* c = x->dev[d].get();
            HIP_TRY(c, hipSetDevice(c->device));
            const int rc = ensure_state(c, kmax, true);
            if (rc) return lift(x, c, rc);
        }
    }
    struct DevJob { std::vector<Source> sub; std::vector<uint8_t> dig; int rc = 0, err_no = 0; int64_t err_src = -1; };
    std::vector<DevJob> job(nd);
    const bool keep_on_device = nd > 1 || (x->flags & SNAPHASH_FLAG_FORCE_GATHER); // digests stay in HBM for the gather
    std::vector<double> gbusy(nd, 0.0);
    auto run_dev = [&](size_t d) {
        const double tg0 = now_ms();
        struct Busy { double& out; double t0; ~Busy() { out = now_ms() - t0; } } busy{gbusy[d], tg0};
        DevJob& J = job[d];
        J.sub.reserve(member[d].size());
        for (uint32_t g : member[d]) J.sub.push_back(src[g]);
        if (!keep_on_device) J.dig.resize(J.sub.size() * 64);
        J.rc = hash_sources(x->dev[d].get(), J.sub, keep_on_device ? nullptr : J.dig.data(), nullptr, &J.err_no, &J.err_src);
    };

    // host part: a pool of threads over the host-assigned streams, longest first
    std::atomic<size_t> hnext{0};
    std::atomic<int> herr{0};
    std::atomic<int64_t> herr_src{-1};
    unsigned nh = hidx.empty() ? 0u : std::max(1u, std::min<unsigned>(plan_threads, (unsigned)hidx.size()));
    const bool gpu_part = !gidx.empty();
    const bool spare_cores = 2u * nh <= (x->cpus_call ? x->cpus_call : x->cpus); // a long file stream may take a reader thread beside its hasher (hostsha.h)
    std::vector<double> hbusy(std::max(1u, nh), 0.0);
    std::stable_sort(hidx.begin(), hidx.end(), [&](uint32_t a, uint32_t b) { return src[a].len > src[b].len; });
    // A thread takes streams off the queue (longest first) and runs up to eight of them side by side, a stream per 64-bit
    // lane (hostsha_x8.cpp: ~3x a core's one-stream rate) -- but a stream that is a quarter of a thread's share or more
    // keeps a core to itself: it sets the makespan, and lanes share the core (config 5's 255 MiB head).
    uint64_t host_bytes_planned = 0;
    for (uint32_t g : hidx) host_bytes_planned += src[g].len;
    const uint64_t alone_from = nh ? host_bytes_planned / nh / 4 : 0;
    size_t n_lane_streams = 0;
    for (uint32_t g : hidx) n_lane_streams += src[g].len < alone_from || alone_from == 0;
    unsigned host_lanes = nh ? (unsigned)std::min<size_t>(8, (n_lane_streams + nh - 1) / nh) : 1u;
    // ONE descriptor budget for the call: a host thread holds a descriptor per lane, the staging fill keeps a
    // file's descriptor between the batches it appears in (FdCache) -- on a host of 128 cores that is 1 000-2 000 lanes'
    // worth where RLIMIT_NOFILE may be 1 024 (SNAPHASH_FLAG_KEEP_RLIMIT, a low hard limit), and the reference's loop
    // holds ONE.  The soft limit less a reserve for the application is divided: the lanes take what they need up to half
    // of it beside a GPU part (all of it alone) -- fewer lanes per thread, then fewer threads -- the kept descriptors get
    // the rest.  An open that still meets EMFILE waits for a descriptor (do_read_op, host_sha512_many).
    for (auto& d : x->dev) d->fd_call_budget = -1;
    if (src[0].path != nullptr) {
        struct rlimit rl;
        int64_t soft = 1024;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0) soft = rl.rlim_cur == RLIM_INFINITY ? 65536 : (int64_t)rl.rlim_cur;
        // what is not ours: the descriptors the process holds right now (+ a few for what it opens meanwhile), and at
        // least the reserve snaphash_init leaves an application (512, or half of a small limit)
        // (counted only where it can matter -- a listing of /proc/self/fd is tens of microseconds, and the one-file call has 200:
        // with the limit snaphash_init leaves, 65 536 as a rule, the application's reserve covers what it holds)
        const int64_t held = soft < 8192 ? open_descriptors() + 16 : 0;
        const int64_t reserve = std::max<int64_t>(held, std::min<int64_t>(512, soft / 2));
        // a fill thread holds one descriptor for the length of a pread when its file's is not a kept one
        const int64_t transient = gpu_part ? (int64_t)std::min(x->d0()->fill_cap, 12u) * (int64_t)nd : 0;
        const int64_t pool = std::max<int64_t>(2, soft - reserve - transient);
        int64_t lanes_fds = (int64_t)nh * host_lanes;
        const int64_t lanes_cap = gpu_part ? std::max<int64_t>(1, pool / 2) : pool;
        if (lanes_fds > lanes_cap) {
            if ((int64_t)nh > lanes_cap) nh = (unsigned)lanes_cap;
            host_lanes = (unsigned)std::max<int64_t>(1, lanes_cap / nh);
            lanes_fds = (int64_t)nh * host_lanes;
        }
        for (auto& d : x->dev) d->fd_call_budget = std::max<int64_t>(0, (pool - lanes_fds) / (int64_t)nd);
    }
    auto run_host = [&](unsigned t) {
        const double t0 = now_ms();
        int64_t bad = -1;
        const int err = host_sha512_many(
            host_lanes,
            [&]() -> int64_t {
                if (herr.load()) return -1;
                const size_t k = hnext.fetch_add(1);
                return k < hidx.size() ? (int64_t)hidx[k] : -1;
            },
            [&](int64_t g) {
                HostStream h;
                h.mem = src[g].mem;
                h.path = src[g].mem ? nullptr : src[g].path;
                h.len = src[g].len;
                h.digest = digests + 64 * (size_t)g;
                h.read_ahead = spare_cores;
                h.alone = alone_from != 0 && src[g].len >= alone_from;
                return h;
            },
            &bad);
        if (err) {
            int z = 0;
            if (herr.compare_exchange_strong(z, err)) herr_src.store(bad);
        }
        hbusy[t] = now_ms() - t0;
    };

    // A call whose streams all went to the host has no GPU part: the calling thread is the first of the pool (the literal
    // one-file helpers.Sha512sum starts no thread at all).  Otherwise the caller drives the engine(s).
    ThreadJoiner th, dth;
    try {
        for (unsigned t = gpu_part ? 0u : 1u; t < nh; ++t) th.spawn(run_host, t);
        if (gpu_part && nd > 1)
            for (size_t d = 0; d < nd; ++d) dth.spawn(run_dev, d);
    } catch (...) { // no thread to be had: end what runs (the joiners wait for it), report, hash nothing further
        int z = 0;
        herr.compare_exchange_strong(z, EAGAIN);
        th.join_all();
        dth.join_all();
        return fail(x, SNAPHASH_ENOMEM, "could not start a worker thread");
    }
    if (!gpu_part && !x->gpu_only) { // (no fill will be measured by this call: planner.h PlanCalib::relax)
        std::lock_guard<std::mutex> lk(x->calib_mu);
        x->calib.relax(src[0].path != nullptr);
    }
    if (!gpu_part) { if (nh) run_host(0); }
    else if (nd == 1) run_dev(0);
    dth.join_all();
    th.join_all();

    for (uint32_t g : hidx) { x->ex.host_bytes += src[g].len; }
    x->ex.host_streams = hidx.size();
    for (double b : hbusy) x->ex.host_ms = std::max(x->ex.host_ms, b);
    for (double b : gbusy) x->ex.gpu_ms = std::max(x->ex.gpu_ms, b);
    x->ex.host_threads_run = nh;
    // what the host threads of THIS box do against what the model said -- unless the call as a whole took three times its plan:
    // whatever happened there (a cold page cache, a neighbour on the box) is not what the next call will meet
    if (!x->gpu_only && x->ex.planned_host_ms > 0 && !herr.load() &&
        now_ms() - t_hash0 <= 3.0 * std::max(x->ex.planned_host_ms, x->ex.planned_gpu_ms)) {
        std::lock_guard<std::mutex> lk(x->calib_mu);
        x->calib.observe_host(x->ex.planned_host_ms * 1e-3, x->ex.host_ms * 1e-3);
    }

    // first error (lowest walk index) fails the call, as the reference's loop would (build.go:242-244)
    int64_t bad = -1;
    int bad_errno = 0, bad_rc = 0;
    DevCtx* bad_dev = nullptr;
    for (size_t d = 0; d < nd; ++d) {
        if (!job[d].rc) continue;
        const int64_t g = job[d].err_src >= 0 ? (int64_t)member[d][job[d].err_src] : (int64_t)n;
        if (bad < 0 || g < bad) { bad = g; bad_errno = job[d].err_no; bad_rc = job[d].rc; bad_dev = x->dev[d].get(); }
    }
    if (herr.load() && (bad < 0 || herr_src.load() < bad)) {
        bad = herr_src.load();
        bad_errno = herr.load();
        bad_rc = SNAPHASH_EIO;
        bad_dev = nullptr;
        x->last_error = std::string(src[bad].path ? src[bad].path : "<buffer>") + ": " + strerror(bad_errno);
    }
    if (bad_rc) {
        if (bad_dev) x->last_error = bad_dev->last_error;
        if (status && bad >= 0 && bad < (int64_t)n) status[bad] = bad_errno;
        merge_stats(x);
        return bad_rc;
    }

    if (!gpu_part) {
        // every stream went to a host thread: the digests are where they belong already, nothing to gather
    } else if (!keep_on_device) {
        for (size_t k = 0; k < member[0].size(); ++k) memcpy(digests + 64 * (size_t)member[0][k], job[0].dig.data() + 64 * k, 64);
    } else {
        std::vector<size_t> cnt(nd);
        size_t kmax = 1;
        for (size_t d = 0; d < nd; ++d) { cnt[d] = member[d].size(); kmax = std::max(kmax, cnt[d]); }
        std::vector<uint8_t> rows;
        int rc = gather_digest_slabs(x, cnt, kmax, rows);
        if (rc) { merge_stats(x); return rc; }
        for (size_t d = 0; d < nd; ++d)
            for (size_t k = 0; k < cnt[d]; ++k) memcpy(digests + 64 * (size_t)member[d][k], rows.data() + (d * kmax + k) * 64, 64);
    }
    merge_stats(x);
    return SNAPHASH_OK;
}

// sizes (may be NULL): the length each path is expected to have (the walk's lstat, build.go:240-252);
// without it the length comes from stat() here.  Either way a file that is shorter or longer when it
// is read fails the call (io.Copy reads to EOF: the record's size and digest must describe the same bytes).
// os.Open follows symlinks (helpers.go:189); a directory opens but its read fails with EISDIR.  The first path
// (in list order) that cannot be hashed, or -1; what the reference's serial loop would have stopped at.
static int64_t first_bad_path(const char* const* paths, size_t n, int* err)
{
    for (size_t i = 0; i < n; ++i) {
        struct stat st;
        int e = 0;
        if (stat(paths[i], &st) != 0) e = errno;
        else if (S_ISDIR(st.st_mode)) e = EISDIR;
        else if (access(paths[i], R_OK) != 0) e = errno;
        if (e) { *err = e; return (int64_t)i; }
    }
    return -1;
}

int hash_paths(snaphash_ctx* x, const char* const* paths, size_t n, const int64_t* sizes, uint8_t* digests, int32_t* status)
{
    // Optimistic: no per-file checks in front (they were two system calls per file, serial: 0.2 s of a 100 000-file
    // pass); the readers meet any error where it is.  Only the length is needed here: the walk's Lstat supplies it
    // (build.go:240-252), otherwise one stat.
    std::vector<Source> src(n);
    for (size_t i = 0; i < n; ++i) {
        if (!paths[i]) return fail(x, SNAPHASH_EINVAL, "NULL path");
        src[i].path = paths[i];
        if (sizes && sizes[i] >= 0) src[i].len = (uint64_t)sizes[i];
    }
    { // the lengths nobody supplied: one stat each, on a few threads when there are many
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(12u, std::max(1u, std::thread::hardware_concurrency())), n / 2048));
        std::vector<int64_t> bad(T, -1);
        auto work = [&](unsigned t) {
            const size_t lo = n * t / T, hi = n * (t + 1) / T;
            for (size_t i = lo; i < hi; ++i) {
                if (sizes && sizes[i] >= 0) continue;
                struct stat st;
                if (stat(paths[i], &st) != 0 || S_ISDIR(st.st_mode)) { bad[t] = (int64_t)i; return; }
                src[i].len = (uint64_t)st.st_size;
            }
        };
        ThreadJoiner th; // (an emplace_back that throws leaves through the entry point's catch; what was started is joined)
        for (unsigned t = 1; t < T; ++t) th.spawn(work, t);
        work(0);
        th.join_all();
        int64_t first = -1;
        for (unsigned t = 0; t < T; ++t) if (bad[t] >= 0 && (first < 0 || bad[t] < first)) first = bad[t];
        if (first >= 0) { // an earlier path may be unreadable: the reference would have stopped there
            int err = EIO;
            int64_t b = first_bad_path(paths, (size_t)first + 1, &err);
            if (b < 0) b = first;
            if (status) { for (size_t k = 0; k < n; ++k) status[k] = 0; status[b] = err; }
            return fail(x, SNAPHASH_EIO, std::string(paths[b]) + ": " + strerror(err));
        }
    }
    const int rc = hash_sources_top(x, src, digests, status);
    if (rc == SNAPHASH_EIO) {
        // Which file a parallel pass trips over first is a matter of timing; the reference's serial loop stops at the
        // first one in order.  Name that one, if an ordered look finds it.
        int err = 0;
        const int64_t bad = first_bad_path(paths, n, &err);
        if (bad >= 0) {
            if (status) { for (size_t k = 0; k < n; ++k) status[k] = 0; status[bad] = err; }
            return fail(x, SNAPHASH_EIO, std::string(paths[bad]) + ": " + strerror(err));
        }
    }
    return rc;
}

void end_top(snaphash_ctx* x, double t0) { x->stats.wall_ms = now_ms() - t0; }

} // namespace

// ================================== C ABI ==========================================

extern "C" {

int snaphash_abi_version(void) { return SNAPHASH_ABI_VERSION; }

static thread_local std::string g_init_error; // why the last snaphash_init on this thread failed

static int init_fail(int code, const std::string& what, hipError_t e = hipSuccess)
try {
    g_init_error = what + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
    return code;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

static void destroy_dev(DevCtx* c)
{
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (Slot& s : c->slot) {
        if (s.h_buf) (void)hipHostFree(s.h_buf);
        if (s.d_buf) (void)hipFree(s.d_buf);
        if (s.h_jobs) (void)hipHostFree(s.h_jobs);
        if (s.d_jobs) (void)hipFree(s.d_jobs);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.copied) (void)hipEventDestroy(s.copied);
    }
    for (SubSlot& q : c->sub) {
        if (q.h_jobs) (void)hipHostFree(q.h_jobs);
        if (q.d_jobs) (void)hipFree(q.d_jobs);
        if (q.done) (void)hipEventDestroy(q.done);
        if (q.copied) (void)hipEventDestroy(q.copied);
    }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->h_chunks) (void)hipHostFree(c->h_chunks);
    if (c->d_chunks) (void)hipFree(c->d_chunks);
    if (c->d_equal) (void)hipFree(c->d_equal);
    if (c->h_jobs) (void)hipHostFree(c->h_jobs);
    if (c->d_jobs) (void)hipFree(c->d_jobs);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_digests) (void)hipFree(c->d_digests);
    if (c->d_zslots) (void)hipFree(c->d_zslots);
    if (c->d_zout) (void)hipFree(c->d_zout);
    if (c->d_ztoks) (void)hipFree(c->d_ztoks);
    if (c->d_zsizes) (void)hipFree(c->d_zsizes);
    if (c->d_zprefix) (void)hipFree(c->d_zprefix);
    if (c->h_zsizes) (void)hipHostFree(c->h_zsizes);
    if (c->h_zprefix) (void)hipHostFree(c->h_zprefix);
    for (uint8_t* z : c->h_zout) if (z) (void)hipHostFree(z);
    if (c->z_stream) (void)hipStreamDestroy(c->z_stream);
    if (c->z2_stream) (void)hipStreamDestroy(c->z2_stream);
    for (hipEvent_t e : c->z_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->z_part_ev) (void)hipEventDestroy(e);
    for (EventPair& p : c->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
}

// The constant of the planner's model that differs most from box to box, measured where the ctx is made (~1 ms): what the
// first engine's copy engine moves over its link -- two timed H2D copies out of pinned memory, a quarter and a whole of
// 8 MiB, so that what a copy costs whatever its size drops out of the difference.  Every staged call corrects it
// afterwards, and measures what a fill thread moves (hash_sources; a probe of ONE thread copying 4 MiB was tried here and
// says 46 GB/s where a thread inside the pipeline moves 12-14: profiles/r05_calibration.txt).  Never fatal: a probe that
// cannot run leaves the model's defaults in place.
static void calibrate_at_init(snaphash_ctx* x)
{
    if (getenv("SNAPHASH_NO_CALIBRATION")) return;
    DevCtx* c = x->d0();
    if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); return; }
    constexpr uint64_t kProbe = 8u << 20;
    if (c->staging < kProbe) return; // (a ctx made with toy staging buffers, as tests do, is not worth a probe)
    if (ensure_slots(c, 1, kProbe) != SNAPHASH_OK) { c->last_error.clear(); (void)hipGetLastError(); return; }
    Slot& s = c->slot[0];
    const size_t big = (size_t)kProbe, small = big / 4;
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    bool ok = true;
    for (hipEvent_t& ev : e) ok = ok && hipEventCreate(&ev) == hipSuccess;
    if (ok) {
        ok = hipMemcpyAsync(s.d_buf, s.h_buf, small, hipMemcpyHostToDevice, c->copy_stream) == hipSuccess; // warm: the first copy of a stream pays for its set-up
        ok = ok && hipEventRecord(e[0], c->copy_stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(s.d_buf, s.h_buf, small, hipMemcpyHostToDevice, c->copy_stream) == hipSuccess;
        ok = ok && hipEventRecord(e[1], c->copy_stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(s.d_buf, s.h_buf, big, hipMemcpyHostToDevice, c->copy_stream) == hipSuccess;
        ok = ok && hipEventRecord(e[2], c->copy_stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(c->copy_stream) == hipSuccess;
        float t_small = 0, t_big = 0;
        ok = ok && hipEventElapsedTime(&t_small, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&t_big, e[1], e[2]) == hipSuccess;
        if (ok && t_big > t_small) x->calib.observe_dma((double)(big - small), (double)(t_big - t_small) * 1e-3);
    }
    for (hipEvent_t ev : e) if (ev) (void)hipEventDestroy(ev);
    if (!ok) (void)hipGetLastError();
}

int snaphash_init(const snaphash_config* cfg, snaphash_ctx** out)
try {
    if (!out) return SNAPHASH_EINVAL;
    *out = nullptr;
    g_init_error.clear();
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return init_fail(SNAPHASH_EDEVICE, "hipGetDeviceCount found no device", e);
    const bool v1 = cfg && cfg->struct_size >= offsetof(snaphash_config, devices);
    const bool v2 = cfg && cfg->struct_size >= sizeof(snaphash_config);
    std::vector<int> devs;
    if (v2 && cfg->n_devices > 0) {
        if (!cfg->devices) return init_fail(SNAPHASH_EINVAL, "n_devices > 0 with a NULL device list");
        if (cfg->n_devices == 1 && cfg->devices[0] == -1) {
            for (int d = 0; d < ndev; ++d) devs.push_back(d); // every visible device
        } else {
            for (uint32_t k = 0; k < cfg->n_devices; ++k) devs.push_back(cfg->devices[k]);
        }
        if (cfg->stream && devs.size() > 1) return init_fail(SNAPHASH_EINVAL, "a caller stream needs a single-device ctx");
    } else if (!cfg && getenv("SNAPHASH_DEVICES")) {
        // A caller that passes no config (the cgo shim's default) can be steered from the environment, as the
        // reference's build has no config file: SNAPHASH_DEVICES = "all" or "0,1,...".
        const char* p = getenv("SNAPHASH_DEVICES");
        if (!strcmp(p, "all") || !strcmp(p, "-1")) {
            for (int d = 0; d < ndev; ++d) devs.push_back(d);
        } else {
            while (*p) {
                char* end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p || (*end && *end != ',')) return init_fail(SNAPHASH_EINVAL, "SNAPHASH_DEVICES: expected \"all\" or a comma-separated list of ordinals");
                devs.push_back((int)d);
                p = *end ? end + 1 : end;
            }
            if (devs.empty()) return init_fail(SNAPHASH_EINVAL, "SNAPHASH_DEVICES is empty");
        }
    } else {
        int dev = v1 ? cfg->device : -1;
        if (dev < 0 && (e = hipGetDevice(&dev)) != hipSuccess) return init_fail(SNAPHASH_EDEVICE, "hipGetDevice", e);
        devs.push_back(dev);
    }
    if (devs.size() > 64) return init_fail(SNAPHASH_EINVAL, "more than 64 engines");
    std::unique_ptr<snaphash_ctx> x(new (std::nothrow) snaphash_ctx());
    if (!x) return SNAPHASH_ENOMEM;
    // Planning (planner.h; snaphash.h, snaphash_config.host_threads): by default the planner may use every core this
    // process may keep busy; an explicit count fixes it; SNAPHASH_FLAG_GPU_ONLY keeps every byte on the GPU.
    const unsigned ncpu = usable_cpus();
    x->cpus = ncpu;
    if (v2) {
        x->flags = cfg->flags;
        x->host_threads = std::min<uint32_t>(cfg->host_threads, 256);
    }
    if (!cfg && getenv("SNAPHASH_HOST_THREADS")) { // likewise for a ctx created without a config: N threads, 0 = every byte on the GPU
        x->host_threads = (uint32_t)std::min<unsigned long>(strtoul(getenv("SNAPHASH_HOST_THREADS"), nullptr, 10), 256);
        if (x->host_threads == 0) x->flags |= SNAPHASH_FLAG_GPU_ONLY;
    }
    x->gpu_only = (x->flags & SNAPHASH_FLAG_GPU_ONLY) != 0;
    if (!x->gpu_only) x->host_rate = measure_host_rate();
    // where the staging memory and its fill threads live: the GPU's own NUMA node (hostfill.h).  SNAPHASH_SYSFS_ROOT
    // points the topology probe at another tree (the tests' fake one).
    const char* sysfs_env = getenv("SNAPHASH_SYSFS_ROOT");
    const std::string sysfs = sysfs_env && *sysfs_env ? sysfs_env : "/sys";
    const bool numa_on = !(x->flags & SNAPHASH_FLAG_NO_NUMA) && numa_node_count(sysfs) > 1;
    const unsigned fill_cap = std::max(2u, std::min(12u, ncpu / (unsigned)devs.size()));
    // Descriptors the engines may keep open between batches (FdCache): what RLIMIT_NOFILE leaves after a reserve for the
    // application, shared among the engines.  The soft limit is raised to the hard one first (at most 65 536), as the Go
    // runtime itself does at start-up since 1.19: a tree of 10 000 files would otherwise be opened segment by segment.
    int64_t fd_budget = 0;
    {
        struct rlimit rl;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0) {
            const rlim_t want = rl.rlim_max == RLIM_INFINITY ? 65536 : std::min<rlim_t>(rl.rlim_max, 65536);
            const bool keep = (cfg && cfg->struct_size >= sizeof(snaphash_config) && (cfg->flags & SNAPHASH_FLAG_KEEP_RLIMIT)) || getenv("SNAPHASH_KEEP_RLIMIT");
            if (!keep && rl.rlim_cur != RLIM_INFINITY && rl.rlim_cur < want) {
                struct rlimit up = rl;
                up.rlim_cur = want;
                if (setrlimit(RLIMIT_NOFILE, &up) == 0) rl = up;
            }
            const int64_t soft = rl.rlim_cur == RLIM_INFINITY ? 65536 : (int64_t)rl.rlim_cur;
            fd_budget = std::max<int64_t>(0, std::min<int64_t>(soft - 512, 32768)) / (int64_t)devs.size();
        }
    }
    for (size_t k = 0; k < devs.size(); ++k) {
        const int dev = devs[k];
        if (dev < 0 || dev >= ndev) {
            for (auto& d : x->dev) destroy_dev(d.get());
            return init_fail(SNAPHASH_EINVAL, "device ordinal out of range");
        }
        hipDeviceProp_t prop;
        int rc = SNAPHASH_OK;
        std::string why;
        if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) { rc = SNAPHASH_EDEVICE; why = "hipGetDeviceProperties"; }
        else if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { // the code object is gfx950-only
            rc = SNAPHASH_EDEVICE;
            e = hipSuccess;
            why = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        } else if ((e = hipSetDevice(dev)) != hipSuccess) { rc = SNAPHASH_EDEVICE; why = "hipSetDevice"; }
        std::unique_ptr<DevCtx> c(rc ? nullptr : new (std::nothrow) DevCtx());
        if (!rc && !c) rc = SNAPHASH_ENOMEM;
        if (!rc) {
            c->device = dev;
            c->index = (int)k;
            if (v1) {
                if (cfg->staging_bytes) c->staging = (cfg->staging_bytes + kAlign - 1) & ~(uint64_t)(kAlign - 1);
                c->kernel_pref = cfg->kernel;
                if (cfg->deflate_depth) c->deflate_depth = std::min<uint32_t>(std::max<uint32_t>(cfg->deflate_depth, 4u), 256u) & ~3u; // whole batches of four links
                if (c->kernel_pref == SNAPHASH_KERNEL_QUAD && !have_quad_kernel()) {
                    rc = SNAPHASH_EINVAL;
                    e = hipSuccess;
                    why = "SNAPHASH_KERNEL_QUAD is not in this build (make QUAD=1)";
                }
                if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
            }
        }
        if (!rc) {
            if (c->staging < (1u << 16)) c->staging = 1u << 16;
            c->fill_cap = fill_cap;
            c->fd_budget = fd_budget;
            if (!cfg && getenv("SNAPHASH_DEFLATE_DEPTH")) // a ctx made without a config is steered from the environment (snaphash.h)
                c->deflate_depth = std::min<uint32_t>(std::max<uint32_t>((uint32_t)strtoul(getenv("SNAPHASH_DEFLATE_DEPTH"), nullptr, 10), 4u), 256u) & ~3u;
            { // SPX: 8; a CPX/DPX partition presents fewer
                int xcc = 0;
                if (hipDeviceGetAttribute(&xcc, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && xcc >= 1 && xcc <= 64) c->n_xcd = (uint32_t)xcc;
                else (void)hipGetLastError();
            }
            std::vector<int> cpus;
            char bdf[64] = {0};
            if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, dev) == hipSuccess) c->pci_bus_id = bdf;
            else (void)hipGetLastError();
            if (numa_on) {
                c->numa_node = numa_node_of_pci(sysfs, c->pci_bus_id);
                cpus = numa_cpus_of_node(sysfs, c->numa_node);
            }
            c->pool.configure(256, cpus); // threads are created as batches ask for them (run_reads caps the count)
            if (!c->stream) {
                if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { rc = SNAPHASH_EDEVICE; why = "hipStreamCreateWithFlags"; }
                else c->own_stream = true;
            }
        }
        if (rc) {
            for (auto& d : x->dev) destroy_dev(d.get());
            return init_fail(rc, why, e);
        }
        x->dev.push_back(std::move(c));
    }
    // engines that share a NUMA node get disjoint slices of its CPUs (hostfill.h slice_cpus)
    if (numa_on && x->dev.size() > 1) {
        for (size_t k = 0; k < x->dev.size(); ++k) {
            DevCtx* c = x->dev[k].get();
            if (c->numa_node < 0) continue;
            size_t m = 0, pos = 0;
            for (size_t j = 0; j < x->dev.size(); ++j)
                if (x->dev[j]->numa_node == c->numa_node) { if (j == k) pos = m; ++m; }
            if (m < 2) continue;
            const std::vector<int> mine = slice_cpus(numa_cpus_of_node(sysfs, c->numa_node), pos, m);
            if (!mine.empty()) c->pool.configure(256, mine);
        }
    }
    for (auto& d : x->dev) { d->owner = x.get(); d->owner_cpus_per_engine = std::max(1u, ncpu / (unsigned)x->dev.size()); }
    if (!x->gpu_only) calibrate_at_init(x.get()); // a ctx that may plan measures the box it plans for (planner.h PlanCalib)
    *out = x.release();
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

void snaphash_destroy(snaphash_ctx* x)
{
    if (!x) return;
    if (x->open_batch) snaphash_batch_abort(x->open_batch);
    for (size_t d = 0; d < x->d_gather.size(); ++d)
        if (x->d_gather[d]) { (void)hipSetDevice(x->dev[d]->device); (void)hipFree(x->d_gather[d]); }
    if (x->rccl.ok)
        for (ncclComm_t c : x->rccl.comms) if (c) (void)x->rccl.CommDestroy(c);
    for (auto& d : x->dev) destroy_dev(d.get());
    delete x;
}

#define TOP_ENTER(x)                                                                  \
    if ((x)->open_batch) return fail((x), SNAPHASH_EINVAL, "a streaming batch is open on this ctx"); \
    for (auto& d_ : (x)->dev) {                                                       \
        (void)hipSetDevice(d_->device);                                               \
        int rc_ = sync_ctx(d_.get());                                                 \
        if (rc_) return lift((x), d_.get(), rc_);                                     \
    }                                                                                 \
    begin_top(x);                                                                     \
    const double t_top0_ = now_ms()

int snaphash_sha512_files(snaphash_ctx* x, const char* const* paths, size_t n, uint8_t* digests, int32_t* status)
try {
    if (!x || (n && (!paths || !digests))) return fail(x, SNAPHASH_EINVAL, "bad argument");
    TOP_ENTER(x);
    int rc = hash_paths(x, paths, n, nullptr, digests, status);
    end_top(x, t_top0_);
    return rc;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_sha512_buffers(snaphash_ctx* x, const void* const* bufs, const uint64_t* lens, size_t n, uint8_t* digests)
try {
    if (!x || (n && (!bufs || !lens || !digests))) return fail(x, SNAPHASH_EINVAL, "bad argument");
    TOP_ENTER(x);
    std::vector<Source> src(n);
    for (size_t i = 0; i < n; ++i) {
        if (!bufs[i] && lens[i]) return fail(x, SNAPHASH_EINVAL, "NULL buffer with non-zero length");
        src[i].mem = bufs[i] ? (const uint8_t*)bufs[i] : (const uint8_t*)"";
        src[i].len = lens[i];
    }
    int rc = hash_sources_top(x, src, digests, nullptr);
    end_top(x, t_top0_);
    return rc;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_sha512_device(snaphash_ctx* x, const void* d_base, const uint64_t* offsets, const uint64_t* lens,
                           size_t n, void* d_digests)
try {
    if (!x || (n && (!d_base || !offsets || !lens || !d_digests))) return fail(x, SNAPHASH_EINVAL, "bad argument");
    if (((uintptr_t)d_base & 15) != 0) return fail(x, SNAPHASH_EINVAL, "d_base must be 16-byte aligned");
    if (n > 0xffffffffull) return fail(x, SNAPHASH_EINVAL, "too many streams");
    if (x->open_batch) return fail(x, SNAPHASH_EINVAL, "a streaming batch is open on this ctx");
    DevCtx* c = x->d0(); // resident data lives on one device: the ctx's first engine
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c); // one call in flight per ctx: the pinned job array is reused
    if (rc) return lift(x, c, rc);
    begin_top(x);
    begin_call(c);
    if (n == 0) return SNAPHASH_OK;
    rc = ensure_state(c, n, false);
    if (rc) return lift(x, c, rc);
    rc = ensure_jobs(c, &c->h_jobs, &c->d_jobs, &c->jobs_cap, n);
    if (rc) return lift(x, c, rc);
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i] & 15) return fail(x, SNAPHASH_EINVAL, "offsets must be 16-byte aligned");
        if (lens[i] >> 35) return fail(x, SNAPHASH_EINVAL, "a resident stream is limited to 32 GiB per call (32-bit block counters)");
        Job& j = c->h_jobs[i];
        j.data = (uint64_t)(uintptr_t)d_base + offsets[i];
        j.nbytes = lens[i];
        j.total_prev = 0;
        j.idx = (uint32_t)i;
        j.flags = kJobFirst | kJobFinal;
        c->stats.bytes_hashed += lens[i];
        c->stats.blocks += padded_blocks(lens[i], true);
    }
    c->stats.streams = n;
    return lift(x, c, launch_jobs(c, c->h_jobs, c->d_jobs, n, (uint8_t*)d_digests));
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_sync(snaphash_ctx* x)
try {
    if (!x) return SNAPHASH_EINVAL;
    for (auto& d : x->dev) {
        HIP_TRY(d.get(), hipSetDevice(d->device));
        int rc = sync_ctx(d.get());
        if (rc) return lift(x, d.get(), rc);
    }
    const double wall = x->d0()->stats.wall_ms;
    merge_stats(x);
    if (x->stats.wall_ms == 0) x->stats.wall_ms = wall;
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

// ---- the pass -----------------------------------------------------------------------

static int tree_impl(snaphash_ctx* x, const char* build_dir, const char* data_tar, const uint8_t* archive_digest,
                     std::string& yaml)
{
    // build.go:222 hashes the archive first; a missing archive fails before the walk
    struct stat st;
    if (data_tar && stat(data_tar, &st) != 0) return fail(x, SNAPHASH_EIO, std::string(data_tar) + ": " + strerror(errno));
    std::vector<Record> recs;
    int en = 0;
    const double tw0 = now_ms();
    int rc = walk_tree(build_dir, recs, &en);
    const double tw1 = now_ms();
    if (rc) return fail(x, rc, rc == SNAPHASH_EIO ? std::string(build_dir) + ": " + strerror(en) : "Unknown file mode");
    if (const size_t b = first_unemittable_name(recs); b < recs.size())
        return fail(x, SNAPHASH_ENAME, "name outside what the YAML emitter restates (yamlscalar.cpp): " + recs[b].name);
    std::vector<const char*> paths;
    std::vector<int64_t> sizes;
    if (data_tar) { paths.push_back(data_tar); sizes.push_back(-1); } // element 0 = the archive, as in the Go batch shape
    for (const Record& r : recs)
        if (r.is_regular) { paths.push_back(r.path.c_str()); sizes.push_back(r.size); } // info.Size() of the walk's Lstat (build.go:240-252)
    std::vector<uint8_t> dig(paths.size() * 64 + 64);
    const double th0 = now_ms();
    // The YAML of a large tree is written WHILE its files are hashed (round 5): everything but the digests is known after
    // the walk, so one background thread writes the document with zeros where the digests go (2 ms for 10 100 records,
    // inside a pass of 190), and what is left behind the last kernel is the hex of the digests (~0.15 ms) instead of the
    // whole emitter (0.7-1.4 ms).  A small tree is written afterwards as before: a thread costs more than its YAML.
    const bool early_yaml = recs.size() >= 2048;
    YamlSkeleton sk;
    int sk_rc = SNAPHASH_OK;
    ThreadJoiner skt;
    if (early_yaml) skt.spawn([&recs, &sk, &sk_rc] { sk_rc = emit_yaml_skeleton(recs, sk, 1); });
    rc = hash_paths(x, paths.data(), paths.size(), sizes.data(), dig.data(), nullptr);
    skt.join_all();
    const double th1 = now_ms();
    if (rc) return rc;
    const uint8_t* arch = data_tar ? dig.data() : archive_digest;
    const uint8_t* files = data_tar ? dig.data() + 64 : dig.data();
    if (early_yaml) {
        rc = sk_rc;
        if (!rc) { yaml_fill_digests(sk, arch, files); yaml.swap(sk.text); }
    } else {
        rc = emit_yaml(recs, arch, files, yaml);
    }
    static const bool trace_tree = getenv("SNAPHASH_TRACE_TREE") != nullptr;
    if (trace_tree)
        fprintf(stderr, "snaphash tree: walk %.1f ms (%zu records), hash %.1f ms (%zu streams), yaml %.1f ms (%zu bytes)\n", tw1 - tw0,
                recs.size(), th1 - th0, paths.size(), now_ms() - th1, yaml.size());
    return rc;
}

static int write_yaml_file(snaphash_ctx* x, const char* build_dir, const std::string& y)
{
    const std::string dir = std::string(build_dir) + "/DEBIAN";
    const std::string path = dir + "/hashes.yaml";
    int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644); // ioutil.WriteFile(..., 0644)
    if (fd < 0) return fail(x, SNAPHASH_EIO, path + ": " + strerror(errno));
    size_t off = 0;
    while (off < y.size()) {
        ssize_t w = write(fd, y.data() + off, y.size() - off);
        if (w < 0) { if (errno == EINTR) continue; int e = errno; close(fd); return fail(x, SNAPHASH_EIO, path + ": " + strerror(e)); }
        off += (size_t)w;
    }
    close(fd);
    return SNAPHASH_OK;
}

int snaphash_tree_ex(snaphash_ctx* x, const char* build_dir, const char* data_tar, const uint8_t* archive_digest,
                     int write_file, char** yaml_out, size_t* yaml_len)
try {
    if (!x || !build_dir || (!data_tar && !archive_digest) || (!write_file && !yaml_out)) return fail(x, SNAPHASH_EINVAL, "bad argument");
    if (yaml_out) *yaml_out = nullptr;
    if (write_file) {
        const std::string dir = std::string(build_dir) + "/DEBIAN";
        (void)mkdir(dir.c_str(), 0755); // os.MkdirAll(debianDir, 0755), error ignored (build.go:219)
    }
    const double t_in = now_ms();
    TOP_ENTER(x);
    std::string y;
    int rc = tree_impl(x, build_dir, data_tar, archive_digest, y);
    const double t_impl = now_ms();
    if (!rc && write_file) rc = write_yaml_file(x, build_dir, y); // nothing is written on error (build.go:260-267)
    end_top(x, t_top0_);
    if (rc) return rc;
    if (yaml_out) {
        char* p = (char*)malloc(y.size() + 1);
        if (!p) return fail(x, SNAPHASH_ENOMEM, "malloc");
        memcpy(p, y.data(), y.size());
        p[y.size()] = 0;
        *yaml_out = p;
    }
    if (yaml_len) *yaml_len = y.size();
    static const bool trace_tree = getenv("SNAPHASH_TRACE_TREE") != nullptr;
    if (trace_tree)
        fprintf(stderr, "snaphash tree_ex: entry %.2f ms, the pass %.2f ms, the caller's copy of the YAML %.2f ms\n", t_top0_ - t_in, t_impl - t_top0_, now_ms() - t_impl);
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_tree(snaphash_ctx* x, const char* build_dir, const char* data_tar, char** yaml_out, size_t* yaml_len)
try {
    if (!x || !build_dir || !data_tar || !yaml_out) return fail(x, SNAPHASH_EINVAL, "bad argument");
    return snaphash_tree_ex(x, build_dir, data_tar, nullptr, 0, yaml_out, yaml_len);
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_write_hashes(snaphash_ctx* x, const char* build_dir, const char* data_tar)
try {
    if (!x || !build_dir || !data_tar) return fail(x, SNAPHASH_EINVAL, "bad argument");
    return snaphash_tree_ex(x, build_dir, data_tar, nullptr, 1, nullptr, nullptr);
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

static int mismatch(snaphash_ctx* c, snaphash_mismatch* m, int kind, const std::string& name)
{
    if (m) {
        m->kind = kind;
        m->reserved = 0;
        snprintf(m->name, sizeof m->name, "%s", name.c_str());
    }
    static const char* const what[] = {"", "missing on disk", "not in hashes.yaml", "size differs", "sha512 differs",
                                       "mode differs", "archive-sha512 differs"};
    return fail(c, SNAPHASH_EMISMATCH, name + ": " + what[kind]);
}

int snaphash_verify(snaphash_ctx* x, const char* inst_dir, const char* data_tar, const char* yaml, size_t yaml_len,
                    snaphash_mismatch* first)
try {
    if (!x || !inst_dir || !yaml) return fail(x, SNAPHASH_EINVAL, "bad argument");
    TOP_ENTER(x);
    ParsedHashes ph;
    int rc = parse_yaml(yaml, yaml_len, ph);
    if (rc) return fail(x, rc, "hashes.yaml: parse error");
    std::vector<Record> recs;
    int en = 0;
    rc = walk_tree(inst_dir, recs, &en);
    if (rc) return fail(x, rc, rc == SNAPHASH_EIO ? std::string(inst_dir) + ": " + strerror(en) : "Unknown file mode");

    // Both lists are in walk order (per-directory byte-wise pre-order), so a
    // merge-style scan finds the first name present on one side only.
    size_t i = 0, j = 0;
    while (i < ph.files.size() && j < recs.size()) {
        if (ph.files[i].name == recs[j].name) { ++i; ++j; continue; }
        // decide which side is "extra": look the yaml name up on disk
        bool on_disk = false;
        for (size_t k = j; k < recs.size(); ++k)
            if (recs[k].name == ph.files[i].name) { on_disk = true; break; }
        return on_disk ? mismatch(x, first, 2, recs[j].name) : mismatch(x, first, 1, ph.files[i].name);
    }
    if (i < ph.files.size()) return mismatch(x, first, 1, ph.files[i].name);
    if (j < recs.size()) return mismatch(x, first, 2, recs[j].name);

    for (size_t k = 0; k < recs.size(); ++k) {
        const ParsedRecord& p = ph.files[k];
        const Record& r = recs[k];
        char a[11], b[11];
        if (mode_string(p.st_mode, a) || mode_string(r.st_mode, b) || memcmp(a, b, 10) != 0) return mismatch(x, first, 5, r.name);
        if (r.is_regular) {
            if (!p.has_size || p.size != r.size) return mismatch(x, first, 3, r.name);
        } else if (p.has_size || !p.sha512_hex.empty()) {
            return mismatch(x, first, 3, r.name);
        }
    }
    // the caller handed over an archive to check: a yaml without a well-formed archive-sha512 cannot vouch for it
    if (data_tar && (!ph.has_archive || ph.archive_hex.size() != 128)) return mismatch(x, first, 6, "archive-sha512");
    std::vector<const char*> paths;
    std::vector<int64_t> sizes;
    std::vector<size_t> owner;
    if (data_tar) { paths.push_back(data_tar); sizes.push_back(-1); owner.push_back((size_t)-1); }
    for (size_t k = 0; k < recs.size(); ++k)
        if (recs[k].is_regular) { paths.push_back(recs[k].path.c_str()); sizes.push_back(recs[k].size); owner.push_back(k); }
    std::vector<uint8_t> dig(paths.size() * 64 + 64);
    rc = hash_paths(x, paths.data(), paths.size(), sizes.data(), dig.data(), nullptr);
    end_top(x, t_top0_);
    if (rc) return rc;
    for (size_t q = 0; q < paths.size(); ++q) {
        if (owner[q] == (size_t)-1) {
            if (!digest_matches_hex(dig.data() + 64 * q, ph.archive_hex)) return mismatch(x, first, 6, "archive-sha512");
        } else if (!digest_matches_hex(dig.data() + 64 * q, ph.files[owner[q]].sha512_hex)) {
            return mismatch(x, first, 4, recs[owner[q]].name);
        }
    }
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

void snaphash_free(void* p) { free(p); }

// ---- ABI 4: the pass, one process per GPU (SURVEY sec. 8e in the form torch.distributed / MPI launch it) -------------
// Every rank walks the same tree and derives the same LPT plan (deterministic, by SHA-512 block count, the archive is
// stream 0); rank r hashes its members into a slab of `rows` digests; the caller all-gathers the slabs (RCCL); any rank
// turns the gathered slabs into hashes.yaml.  plan and emit are host-only.
struct snaphash_shard {
    std::vector<Record> recs;
    std::string tar_path;
    std::vector<const char*> all_paths;       // stream i of the whole job: [0] = the archive, then the regular files in walk order (into recs / tar_path)
    std::vector<int64_t> all_sizes;
    std::vector<int32_t> shard_of;            // per stream
    std::vector<uint32_t> row_of;             // per stream: its row in its rank's slab
    std::vector<uint32_t> mine;               // streams of this rank, in row order
    std::vector<const char*> my_paths;
    std::vector<int64_t> my_sizes;
    uint32_t rank = 0, world = 1;
    uint32_t local_ranks = 0; // ranks that share this node's cores, as the caller said (0 = not said)
    size_t rows = 1;
    uint64_t my_bytes = 0;
    // hashes.yaml less its digests, written by a background thread from the end of the plan on (a large tree): behind the
    // all-gather rank 0 only has the digests' hex left to write
    YamlSkeleton sk;
    int sk_rc = SNAPHASH_OK;
    bool sk_used = false;
    std::thread sk_thread;
    ~snaphash_shard() { if (sk_thread.joinable()) sk_thread.join(); }
};

// What follows the walk in a rank's plan, whoever walked: names the emitter can write, the stream list (archive first),
// the LPT shares, this rank's members, and the thread that starts writing hashes.yaml.
static int finish_shard_plan(std::unique_ptr<snaphash_shard>& sh, const char* data_tar, int64_t tar_size)
{
    const uint32_t rank = sh->rank, world = sh->world;
    if (first_unemittable_name(sh->recs) < sh->recs.size()) return SNAPHASH_ENAME;
    sh->tar_path = data_tar;
    sh->all_paths.reserve(sh->recs.size() + 1);
    sh->all_sizes.reserve(sh->recs.size() + 1);
    sh->all_paths.push_back(sh->tar_path.c_str());
    sh->all_sizes.push_back(tar_size);
    for (const Record& r : sh->recs) // (recs is not touched again: the pointers stay good)
        if (r.is_regular) { sh->all_paths.push_back(r.path.c_str()); sh->all_sizes.push_back(r.size); }
    const size_t n = sh->all_paths.size();
    std::vector<uint64_t> lens(n);
    for (size_t i = 0; i < n; ++i) lens[i] = (uint64_t)sh->all_sizes[i];
    sh->shard_of.assign(n, 0);
    if (world > 1) lpt_assign(lens.data(), n, (int)world, sh->shard_of.data());
    std::vector<uint32_t> count(world, 0);
    sh->row_of.resize(n);
    for (size_t i = 0; i < n; ++i) sh->row_of[i] = count[sh->shard_of[i]]++;
    sh->rows = 1;
    for (uint32_t c : count) sh->rows = std::max<size_t>(sh->rows, c);
    for (size_t i = 0; i < n; ++i)
        if ((uint32_t)sh->shard_of[i] == rank) {
            sh->mine.push_back((uint32_t)i);
            sh->my_paths.push_back(sh->all_paths[i]);
            sh->my_sizes.push_back(i == 0 ? -1 : sh->all_sizes[i]); // the archive's length is taken when it is read, like snaphash_tree
            sh->my_bytes += lens[i];
        }
    if (sh->recs.size() >= 2048) { // (recs is not touched again; a thread that cannot be had just means the YAML is written at emit)
        snaphash_shard* p = sh.get();
        try {
            sh->sk_thread = std::thread([p] {
                try { p->sk_rc = emit_yaml_skeleton(p->recs, p->sk, 1); } catch (...) { p->sk_rc = SNAPHASH_ENOMEM; }
            });
            sh->sk_used = true;
        } catch (...) { sh->sk_used = false; }
    }
    return SNAPHASH_OK;
}

int snaphash_shard_plan(const char* build_dir, const char* data_tar, uint32_t rank, uint32_t world, snaphash_shard** out)
try {
    if (!build_dir || !data_tar || !out || world == 0 || rank >= world || world > 4096) return SNAPHASH_EINVAL;
    *out = nullptr;
    struct stat st;
    if (stat(data_tar, &st) != 0) return SNAPHASH_EIO; // build.go:222: a missing archive fails before the walk
    std::unique_ptr<snaphash_shard> sh(new snaphash_shard());
    sh->rank = rank;
    sh->world = world;
    int en = 0;
    const double tp0 = now_ms();
    int rc = walk_tree(build_dir, sh->recs, &en);
    if (rc) { errno = en; return rc; }
    const double tp1 = now_ms();
    rc = finish_shard_plan(sh, data_tar, (int64_t)st.st_size);
    if (rc) return rc;
    static const bool trace_tree = getenv("SNAPHASH_TRACE_TREE") != nullptr;
    if (trace_tree)
        fprintf(stderr, "snaphash shard plan: walk %.2f ms (%zu records), names + lists + LPT over %u ranks %.2f ms\n", tp1 - tp0, sh->recs.size(), world, now_ms() - tp1);
    *out = sh.release();
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

// ---- ABI 5: the ranks SHARE the walk ----------------------------------------------------------------------------------
// snaphash_shard_plan has every rank walk the whole tree: 2.8-3.2 ms of a 30 ms step at N = 8, the same Lstat issued eight
// times over, by 128 threads at the same dentries.  Here rank r lists the root (everybody does: one directory), walks the
// subtrees of the root's entries i with i mod world == r (filepath.Walk's order inside each, as walk.h gives it) and writes
// what it found into a blob; the caller all-gathers the blobs (two small collectives: lengths, then bytes) and every
// rank rebuilds the SAME record list from them -- entry i's records come from rank i mod world's blob, in the root's
// sorted order, which is Walk's order -- and goes on as snaphash_shard_plan does.  Anything under a root entry whose name
// begins with "DEBIAN" is never a record (build.go:229 returns before it looks at anything) and is not walked at all.
int snaphash_shard_list(const char* build_dir, uint32_t rank, uint32_t world, void** blob_out, size_t* blob_len)
try {
    if (!build_dir || !blob_out || !blob_len || world == 0 || rank >= world || world > 4096) return SNAPHASH_EINVAL;
    *blob_out = nullptr;
    *blob_len = 0;
    std::string blob;
    int en = 0;
    const int rc = shard_listing(build_dir, rank, world, blob, &en); // hostpass.cpp
    if (rc) { errno = en; return rc; }
    void* p = malloc(blob.size() ? blob.size() : 1);
    if (!p) return SNAPHASH_ENOMEM;
    memcpy(p, blob.data(), blob.size());
    *blob_out = p;
    *blob_len = blob.size();
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

int snaphash_shard_plan_from(const char* build_dir, const char* data_tar, uint32_t rank, uint32_t world, const void* const* blobs,
                             const size_t* blob_lens, snaphash_shard** out)
try {
    if (!build_dir || !data_tar || !blobs || !blob_lens || !out || world == 0 || rank >= world || world > 4096) return SNAPHASH_EINVAL;
    *out = nullptr;
    struct stat st;
    if (stat(data_tar, &st) != 0) return SNAPHASH_EIO; // build.go:222: a missing archive fails before the walk
    const double tp0 = now_ms();
    std::unique_ptr<snaphash_shard> sh(new snaphash_shard());
    sh->rank = rank;
    sh->world = world;
    int rc = records_from_listings(build_dir, world, blobs, blob_lens, sh->recs); // hostpass.cpp
    if (rc) return rc;
    const double tp1 = now_ms();
    rc = finish_shard_plan(sh, data_tar, (int64_t)st.st_size);
    if (rc) return rc;
    static const bool trace_tree = getenv("SNAPHASH_TRACE_TREE") != nullptr;
    if (trace_tree)
        fprintf(stderr, "snaphash shard plan from %u ranks' listings: records %.2f ms (%zu), names + lists + LPT %.2f ms\n", world, tp1 - tp0, sh->recs.size(), now_ms() - tp1);
    *out = sh.release();
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

size_t snaphash_shard_rows(const snaphash_shard* sh) { return sh ? sh->rows : 0; }
size_t snaphash_shard_count(const snaphash_shard* sh) { return sh ? sh->mine.size() : 0; }
uint64_t snaphash_shard_bytes(const snaphash_shard* sh) { return sh ? sh->my_bytes : 0; }
size_t snaphash_shard_streams(const snaphash_shard* sh) { return sh ? sh->all_paths.size() : 0; }
const char* snaphash_shard_path(const snaphash_shard* sh, size_t k) { return sh && k < sh->my_paths.size() ? sh->my_paths[k] : nullptr; }

int snaphash_shard_hash(snaphash_ctx* x, snaphash_shard* sh, uint8_t* slab)
try {
    if (!x || !sh || !slab) return fail(x, SNAPHASH_EINVAL, "bad argument");
    TOP_ENTER(x);
    memset(slab, 0, sh->rows * 64);
    // The ranks of a node share its cores: this process sees the whole job's allowance (affinity mask, cgroup quota), and
    // eight ranks that each planned host threads for all of it would be eight times too many.  A rank plans with its
    // share -- the allowance over the ranks that can be on this node (at most one per visible GPU).
    // How many that is: what the caller said (snaphash_shard_set_local_ranks), else what the launcher says
    // (LOCAL_WORLD_SIZE of torch.distributed.run, Open MPI's OMPI_COMM_WORLD_LOCAL_SIZE), and only then a guess from the
    // visible GPUs -- a launcher that shows every rank ONE device would make that guess 1, and every rank would plan for
    // the whole node.
    unsigned ranks_here = sh->local_ranks;
    for (const char* name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"}) {
        if (ranks_here) break;
        const char* e = getenv(name);
        const long v = e ? strtol(e, nullptr, 10) : 0;
        if (v >= 1 && v <= 4096) ranks_here = (unsigned)v;
    }
    if (!ranks_here) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
        ranks_here = std::min<unsigned>(sh->world, (unsigned)ndev);
    }
    ranks_here = std::max(1u, std::min(ranks_here, sh->world));
    struct Share {
        snaphash_ctx* x;
        ~Share()
        {
            x->cpus_call = 0;
            for (auto& d : x->dev) d->fill_call_cap = 0;
        }
    } share{x};
    if (ranks_here > 1) {
        x->cpus_call = std::max(1u, x->cpus / ranks_here);
        for (auto& d : x->dev) d->fill_call_cap = std::max(2u, x->cpus_call > 1u ? x->cpus_call - 1u : 1u); // (and its fill threads with it)
    }
    const int rc = hash_paths(x, sh->my_paths.data(), sh->my_paths.size(), sh->my_sizes.data(), slab, nullptr);
    end_top(x, t_top0_);
    return rc;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

int snaphash_shard_emit(const snaphash_shard* sh_c, const uint8_t* slabs, char** yaml_out, size_t* yaml_len)
try {
    if (!sh_c || !slabs || !yaml_out) return SNAPHASH_EINVAL;
    snaphash_shard* sh = const_cast<snaphash_shard*>(sh_c); // (the skeleton's thread is joined here: the handle is not shared between threads, snaphash.h)
    *yaml_out = nullptr;
    const size_t n = sh->all_paths.size();
    std::vector<uint8_t> dig(n * 64);
    for (size_t i = 0; i < n; ++i)
        memcpy(dig.data() + 64 * i, slabs + ((size_t)sh->shard_of[i] * sh->rows + sh->row_of[i]) * 64, 64);
    std::string y;
    int rc;
    if (sh->sk_used) {
        if (sh->sk_thread.joinable()) sh->sk_thread.join();
        rc = sh->sk_rc;
        if (!rc) {
            YamlSkeleton filled = sh->sk; // (emit may be called again with other slabs: the skeleton stays as it is)
            yaml_fill_digests(filled, dig.data(), dig.data() + 64);
            y.swap(filled.text);
        }
    } else {
        rc = emit_yaml(sh->recs, dig.data(), dig.data() + 64, y);
    }
    if (rc) return rc;
    char* p = (char*)malloc(y.size() + 1);
    if (!p) return SNAPHASH_ENOMEM;
    memcpy(p, y.data(), y.size());
    p[y.size()] = 0;
    *yaml_out = p;
    if (yaml_len) *yaml_len = y.size();
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

void snaphash_shard_free(snaphash_shard* sh) { delete sh; }

int snaphash_shard_set_local_ranks(snaphash_shard* sh, uint32_t ranks_on_this_node)
{
    if (!sh || ranks_on_this_node > sh->world) return SNAPHASH_EINVAL;
    sh->local_ranks = ranks_on_this_node;
    return SNAPHASH_OK;
}

// A 64-bit hash of what every rank must agree on before the collective: the records as walked (name, mode, size), the
// archive's size, the world and who hashes which stream into which row.  Eight bytes a step (multiply, fold): the
// byte-at-a-time FNV-1a this began as cost a millisecond of every rank's 3 ms plan on the 10 100 records of config 2.
uint64_t snaphash_shard_fingerprint(const snaphash_shard* sh)
{
    if (!sh) return 0;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    auto mix64 = [&h](uint64_t v) {
        h = (h ^ v) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    };
    auto mix = [&mix64](const char* p, size_t n) {
        mix64(n);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); mix64(w); }
        if (i < n) { uint64_t w = 0; memcpy(&w, p + i, n - i); mix64(w); }
    };
    mix64(sh->world);
    mix64(sh->rows);
    mix64(sh->recs.size());
    for (const Record& r : sh->recs) {
        mix(r.name.data(), r.name.size());
        mix64(((uint64_t)r.st_mode << 1) | (r.is_regular ? 1u : 0u));
        mix64(r.is_regular ? (uint64_t)r.size : 0);
    }
    for (size_t i = 0; i < sh->all_sizes.size(); ++i) {
        mix64((uint64_t)sh->all_sizes[i]);
        mix64(((uint64_t)(uint32_t)sh->shard_of[i] << 32) | sh->row_of[i]);
    }
    return h ? h : 1;
}

// ---- ABI 4: the plan of a call (host-only) ---------------------------------------------------------------------
int snaphash_plan_streams(const uint64_t* lens, size_t n, snaphash_plan_model* pm, uint8_t* on_host)
try {
    constexpr uint32_t kAbi4Size = (uint32_t)offsetof(snaphash_plan_model, fill_rate);
    if (!pm || pm->struct_size < kAbi4Size || (n && !lens)) return SNAPHASH_EINVAL;
    PlanModel m;
    if (pm->struct_size >= offsetof(snaphash_plan_model, fill_per_file) && pm->fill_rate > 0) m.fill_rate = pm->fill_rate;
    if (pm->struct_size >= sizeof(snaphash_plan_model) && pm->fill_per_file > 0) m.fill_per_stream = pm->fill_per_file;
    m.n_devices = pm->n_devices ? pm->n_devices : 1;
    m.cpus = pm->cpus ? pm->cpus : usable_cpus();
    m.from_files = pm->from_files != 0;
    m.fill_threads = pm->fill_threads ? pm->fill_threads : (m.from_files ? 12u : 6u);
    m.host_threads = pm->host_threads;
    if (pm->host_rate > 0) m.host_rate = pm->host_rate;
    if (pm->gpu_stream_rate > 0) m.gpu_pair_rate = pm->gpu_stream_rate;
    if (pm->gpu_link > 0) m.gpu_link = pm->gpu_link;
    if (pm->gpu_latency > 0) m.gpu_latency = pm->gpu_latency;
    if (pm->host_lane_gain_pct > 100) m.host_lane_gain = pm->host_lane_gain_pct / 100.0;
    const PlanResult r = plan_streams(lens, n, m);
    if (on_host && n) memcpy(on_host, r.on_host.data(), n);
    pm->gpu_seconds = r.gpu_seconds;
    pm->host_seconds = r.host_seconds;
    pm->host_streams = r.host_streams;
    pm->host_bytes = r.host_bytes;
    pm->host_threads_used = r.host_threads;
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

int snaphash_get_plan_model(const snaphash_ctx* x, int from_files, snaphash_plan_model* pm)
try {
    constexpr uint32_t kAbi4Size = (uint32_t)offsetof(snaphash_plan_model, fill_rate);
    if (!x || !pm || pm->struct_size < kAbi4Size) return SNAPHASH_EINVAL;
    const uint32_t have = std::min<uint32_t>(pm->struct_size, (uint32_t)sizeof(snaphash_plan_model));
    const PlanModel m = plan_model_of(x, from_files != 0);
    snaphash_plan_model v;
    memset(&v, 0, sizeof v);
    v.struct_size = have;
    v.n_devices = m.n_devices;
    v.cpus = m.cpus;
    v.fill_threads = m.fill_threads;
    v.host_threads = m.host_threads;
    v.from_files = m.from_files ? 1u : 0u;
    v.host_rate = m.host_rate;
    v.gpu_stream_rate = m.gpu_pair_rate;
    v.gpu_link = m.gpu_link > 0 ? m.gpu_link : (m.from_files ? 54e9 : 55e9);
    v.gpu_latency = m.gpu_latency;
    v.host_lane_gain_pct = (uint32_t)(m.host_lane_gain * 100.0 + 0.5);
    v.fill_rate = m.fill_rate > 0 ? m.fill_rate : (m.from_files ? 6.5e9 : 9e9);
    v.fill_per_file = m.fill_per_stream > 0 ? m.fill_per_stream : (m.from_files ? 10e-6 : 0.3e-6);
    memcpy(pm, &v, have);
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

static PlanCalib calib_of(const snaphash_plan_calib* c)
{
    PlanCalib k;
    k.dma = c->dma; k.fill_mem = c->fill_mem; k.fill_files = c->fill_files;
    k.n_dma = c->n_dma; k.n_fill_mem = c->n_fill_mem; k.n_fill_files = c->n_fill_files;
    k.host_gain = c->host_gain; k.n_host = c->n_host;
    k.fill_per_file = c->fill_per_file; k.n_fill_per_file = c->n_fill_per_file;
    return k;
}
static void calib_to(const PlanCalib& k, snaphash_plan_calib* c)
{
    c->dma = k.dma; c->fill_mem = k.fill_mem; c->fill_files = k.fill_files;
    c->n_dma = k.n_dma; c->n_fill_mem = k.n_fill_mem; c->n_fill_files = k.n_fill_files;
    c->host_gain = k.host_gain; c->n_host = k.n_host;
    c->fill_per_file = k.fill_per_file; c->n_fill_per_file = k.n_fill_per_file;
}

int snaphash_calib_observe(snaphash_plan_calib* c, int what, double bytes, double seconds)
{
    if (!c || c->struct_size < sizeof(snaphash_plan_calib) || what < 0 || what > 6) return SNAPHASH_EINVAL;
    PlanCalib k = calib_of(c);
    bool took = true;
    if (what == 6) took = k.observe_fill_per_file(bytes, seconds);
    else if (what == 5) took = k.observe_host(bytes, seconds);
    else if (what >= 3) k.relax(what == 4);
    else took = what == 0 ? k.observe_dma(bytes, seconds) : k.observe_fill(what == 2, bytes, seconds);
    calib_to(k, c);
    return took ? 1 : 0;
}

int snaphash_calib_observe_call(snaphash_plan_calib* c, int from_files, double bytes, double streams, double copies, double h2d_seconds,
                                double fill_thread_seconds)
{
    if (!c || c->struct_size < sizeof(snaphash_plan_calib)) return SNAPHASH_EINVAL;
    PlanCalib k = calib_of(c);
    k.observe_call(from_files != 0, bytes, streams, copies, h2d_seconds, fill_thread_seconds, true);
    calib_to(k, c);
    return SNAPHASH_OK;
}

int snaphash_calib_apply(const snaphash_plan_calib* c, snaphash_plan_model* pm)
{
    if (!c || c->struct_size < sizeof(snaphash_plan_calib) || !pm || pm->struct_size < sizeof(snaphash_plan_model)) return SNAPHASH_EINVAL;
    PlanModel m;
    m.from_files = pm->from_files != 0;
    m.gpu_link = pm->gpu_link;
    m.fill_rate = pm->fill_rate;
    m.fill_per_stream = pm->fill_per_file;
    if (pm->host_rate > 0) m.host_rate = pm->host_rate;
    calib_of(c).apply(m);
    pm->gpu_link = m.gpu_link;
    pm->fill_rate = m.fill_rate;
    pm->fill_per_file = m.fill_per_stream;
    pm->host_rate = m.host_rate; // (the model's default, 1.4e9, where the caller named none: corrected by what host parts took)
    return SNAPHASH_OK;
}

int snaphash_get_calib(const snaphash_ctx* x, snaphash_plan_calib* out)
{
    if (!x || !out || out->struct_size < sizeof(snaphash_plan_calib)) return SNAPHASH_EINVAL;
    std::lock_guard<std::mutex> lk(const_cast<snaphash_ctx*>(x)->calib_mu);
    calib_to(x->calib, out);
    return SNAPHASH_OK;
}

uint32_t snaphash_usable_cpus(void) { return usable_cpus(); }

uint32_t snaphash_cgroup_cpu_quota(const char* cgroup_root, const char* proc_self_cgroup)
try {
    if (!cgroup_root || !proc_self_cgroup) return 0;
    return cgroup_cpu_quota(cgroup_root, proc_self_cgroup);
} catch (...) {
    return 0;
}

// ---- streaming batch (row f2): bytes are fed as another pass reads them -----------------------

} // extern "C"

struct snaphash_batch {
    snaphash_ctx* x = nullptr;
    DevCtx* c = nullptr;
    size_t n = 0;
    struct St {
        uint64_t done = 0;    // bytes already placed into jobs (a multiple of 128 until the final segment)
        int64_t job = -1;     // index of this stream's job in the slot being filled
        uint32_t ntail = 0;
        bool ended = false, finished = false;
        uint8_t tail[128];
    };
    std::vector<St> st;
    std::vector<uint32_t> in_slot; // streams that own a job in the slot being filled
    unsigned batch = 0;           // slot index parity
    uint64_t used = 0;            // bytes of the current slot in use
    size_t nj = 0;                // jobs in the current slot
    bool failed = false;
    double t0 = 0;
};

namespace {

Slot& cur_slot(snaphash_batch* b) { return b->c->slot[b->batch & 1]; }

int batch_flush(snaphash_batch* b)
{
    DevCtx* c = b->c;
    Slot& sl = cur_slot(b);
    if (b->nj == 0) return SNAPHASH_OK;
    if (b->used) {
        EventPair* ev = next_events(c, 1);
        if (!ev) return fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
        HIP_TRY(c, hipEventRecord(ev->a, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(sl.d_buf, sl.h_buf, b->used, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(ev->b, c->copy_stream));
    }
    int rc = launch_jobs(c, sl.h_jobs, sl.d_jobs, b->nj, c->d_digests, sl.copied);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(sl.done, c->stream));
    sl.busy = true;
    for (uint32_t s : b->in_slot) b->st[s].job = -1;
    b->in_slot.clear();
    ++b->batch;
    b->used = 0;
    b->nj = 0;
    Slot& nx = cur_slot(b);
    if (nx.busy) { HIP_TRY(c, hipEventSynchronize(nx.done)); nx.busy = false; }
    return ensure_jobs(c, &nx.h_jobs, &nx.d_jobs, &nx.jobs_cap, std::max<size_t>(b->n, 1024));
}

// Places `n` bytes (p1[0..n1) then p2[0..n-n1)) of stream s into staging as (part of) a job.
// n is a multiple of 128 unless final.  May flush.
int batch_place(snaphash_batch* b, size_t s, const uint8_t* p1, size_t n1, const uint8_t* p2, size_t n2, bool final)
{
    DevCtx* c = b->c;
    snaphash_batch::St& st = b->st[s];
    const uint64_t S = c->staging;
    size_t left1 = n1, left2 = n2;
    for (;;) {
        Slot& sl = cur_slot(b);
        const size_t left = left1 + left2;
        bool extend = false;
        if (st.job >= 0) {
            const Job& j = sl.h_jobs[st.job];
            const uint64_t end = (j.data - (uint64_t)(uintptr_t)sl.d_buf) + j.nbytes;
            if ((size_t)st.job == b->nj - 1 && end == b->used) extend = true; // contiguous with its own last bytes
            else { int rc = batch_flush(b); if (rc) return rc; continue; }   // one segment per stream per launch
        }
        uint64_t at = extend ? b->used : ((b->used + kAlign - 1) & ~(uint64_t)(kAlign - 1));
        uint64_t room = at < S ? S - at : 0;
        uint64_t take = left;
        if (take > room) take = room & ~(uint64_t)127;
        if (take == 0 && left > 0) { // slot full
            int rc = batch_flush(b);
            if (rc) return rc;
            continue;
        }
        if (!extend && b->nj >= cur_slot(b).jobs_cap) { int rc = batch_flush(b); if (rc) return rc; continue; }
        const bool fin = final && take == left;
        uint8_t* dst = sl.h_buf + at;
        size_t t1 = std::min<size_t>(left1, take), t2 = (size_t)take - t1;
        if (t1) memcpy(dst, p1 + (n1 - left1), t1);
        if (t2) memcpy(dst + t1, p2 + (n2 - left2), t2);
        left1 -= t1;
        left2 -= t2;
        if (extend) {
            Job& j = sl.h_jobs[st.job];
            j.nbytes += take;
            if (fin) j.flags |= kJobFinal;
        } else {
            Job j;
            j.data = (uint64_t)(uintptr_t)(sl.d_buf + at);
            j.nbytes = take;
            j.total_prev = st.done;
            j.idx = (uint32_t)s;
            j.flags = (st.done == 0 ? kJobFirst : 0u) | (fin ? kJobFinal : 0u);
            st.job = (int64_t)b->nj;
            sl.h_jobs[b->nj++] = j;
            b->in_slot.push_back((uint32_t)s);
        }
        c->stats.blocks += (take >> 7) + (fin ? padded_blocks(take & 127, true) : 0);
        b->used = at + take;
        st.done += take;
        c->stats.bytes_hashed += take;
        if (left1 + left2 == 0) return SNAPHASH_OK;
    }
}

} // namespace

extern "C" {

int snaphash_batch_begin(snaphash_ctx* x, size_t n_streams, snaphash_batch** out)
try {
    if (!x || !out) return fail(x, SNAPHASH_EINVAL, "bad argument");
    *out = nullptr;
    if (n_streams > 0xffffffffull) return fail(x, SNAPHASH_EINVAL, "too many streams");
    TOP_ENTER(x);
    DevCtx* c = x->d0();
    int rc = ensure_slots(c);
    if (!rc) rc = ensure_state(c, std::max<size_t>(n_streams, 1), true);
    if (!rc) rc = ensure_jobs(c, &c->slot[0].h_jobs, &c->slot[0].d_jobs, &c->slot[0].jobs_cap, std::max<size_t>(n_streams, 1024));
    if (rc) return lift(x, c, rc);
    snaphash_batch* b = new (std::nothrow) snaphash_batch();
    if (!b) return fail(x, SNAPHASH_ENOMEM, "new");
    b->x = x;
    b->c = c;
    b->n = n_streams;
    b->st.resize(n_streams);
    b->t0 = t_top0_;
    c->slot[0].busy = c->slot[1].busy = false;
    x->open_batch = b;
    *out = b;
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_batch_append(snaphash_batch* b, size_t stream, const void* data, size_t n)
try {
    if (!b) return SNAPHASH_EINVAL;
    if (b->failed) return fail(b->x, SNAPHASH_EINVAL, "the batch has failed");
    if (stream >= b->n || (!data && n) || b->st[stream].ended) return fail(b->x, SNAPHASH_EINVAL, "bad stream or buffer");
    if (n == 0) return SNAPHASH_OK;
    snaphash_batch::St& st = b->st[stream];
    const uint8_t* p = (const uint8_t*)data;
    if (st.ntail + n < 128) { // still less than a block: keep it on the host
        memcpy(st.tail + st.ntail, p, n);
        st.ntail += (uint32_t)n;
        return SNAPHASH_OK;
    }
    const size_t whole = (st.ntail + n) & ~(size_t)127; // bytes that go to the GPU now
    const size_t from_data = whole - st.ntail;
    HIP_TRY(b->c, hipSetDevice(b->c->device));
    int rc = batch_place(b, stream, st.tail, st.ntail, p, from_data, false);
    if (rc) { b->failed = true; return lift(b->x, b->c, rc); }
    st.ntail = (uint32_t)(n - from_data);
    if (st.ntail) memcpy(st.tail, p + from_data, st.ntail);
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_batch_end(snaphash_batch* b, size_t stream)
try {
    if (!b) return SNAPHASH_EINVAL;
    if (b->failed) return fail(b->x, SNAPHASH_EINVAL, "the batch has failed");
    if (stream >= b->n) return fail(b->x, SNAPHASH_EINVAL, "bad stream");
    snaphash_batch::St& st = b->st[stream];
    if (st.ended) return SNAPHASH_OK;
    st.ended = true;
    HIP_TRY(b->c, hipSetDevice(b->c->device));
    int rc = batch_place(b, stream, st.tail, st.ntail, nullptr, 0, true); // the final (< 128 byte, maybe empty) segment
    if (rc) { b->failed = true; return lift(b->x, b->c, rc); }
    st.ntail = 0;
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_batch_finish(snaphash_batch* b, uint8_t* digests)
try {
    if (!b) return SNAPHASH_EINVAL;
    snaphash_ctx* x = b->x;
    DevCtx* c = b->c;
    int rc = (b->n && !digests) ? fail(x, SNAPHASH_EINVAL, "NULL digests") : SNAPHASH_OK;
    if (!rc && b->failed) rc = fail(x, SNAPHASH_EINVAL, "the batch has failed");
    (void)hipSetDevice(c->device);
    for (size_t s = 0; s < b->n && !rc; ++s)
        if (!b->st[s].ended) rc = snaphash_batch_end(b, s);
    if (!rc) rc = lift(x, c, batch_flush(b));
    if (!rc) rc = lift(x, c, sync_ctx(c));
    c->slot[0].busy = c->slot[1].busy = false;
    if (!rc && b->n && hipMemcpy(digests, c->d_digests, b->n * 64, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(x, SNAPHASH_EDEVICE, "D2H of the digests failed");
    c->stats.streams = b->n;
    merge_stats(x);
    x->stats.wall_ms = now_ms() - b->t0;
    x->open_batch = nullptr;
    delete b;
    return rc;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

void snaphash_batch_abort(snaphash_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->c->device);
    (void)hipStreamSynchronize(b->c->copy_stream);
    (void)hipStreamSynchronize(b->c->stream);
    collect_events(b->c);
    b->c->pending = false;
    b->c->slot[0].busy = b->c->slot[1].busy = false;
    b->x->open_batch = nullptr;
    delete b;
}

} // extern "C"

#include "targz.inc"

// ---- helpers.FilesAreEqual / DirUpdated (row f4) ----------------------------------------------

namespace {

// Chunk tables come in two halves (pinned host + device), so that the table of batch k+1 can be built
// and uploaded while the kernel of batch k still reads its own.
int ensure_chunks(DevCtx* c, size_t n)
{
    if (n <= c->chunks_cap) return SNAPHASH_OK;
    const size_t want = std::max<size_t>(n, 4096);
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // nothing may still read the old tables
    if (c->h_chunks) (void)hipHostFree(c->h_chunks);
    if (c->d_chunks) (void)hipFree(c->d_chunks);
    c->h_chunks = nullptr; c->d_chunks = nullptr; c->chunks_cap = 0;
    HIP_TRY(c, hipHostMalloc((void**)&c->h_chunks, 2 * want * sizeof(CmpChunk), hipHostMallocDefault));
    HIP_TRY(c, hipMalloc((void**)&c->d_chunks, 2 * want * sizeof(CmpChunk)));
    c->chunks_cap = want;
    return SNAPHASH_OK;
}

// Ranges (device addresses) -> chunk table -> kernel.  d_equal must hold one byte per pair.
int launch_compare_ranges(DevCtx* c, const std::vector<uint64_t>& a, const std::vector<uint64_t>& b,
                          const std::vector<uint64_t>& lens, uint8_t* d_equal, int half = 0)
{
    const size_t n = lens.size();
    size_t nchunks = 0;
    for (size_t i = 0; i < n; ++i) nchunks += (size_t)((lens[i] + kCmpChunk - 1) / kCmpChunk);
    if (nchunks > 0xffffffffull) return fail(c, SNAPHASH_EINVAL, "too many comparison chunks");
    int rc = ensure_chunks(c, nchunks);
    if (rc) return rc;
    CmpChunk* h_tab = c->h_chunks + (size_t)half * c->chunks_cap;
    CmpChunk* d_tab = c->d_chunks + (size_t)half * c->chunks_cap;
    size_t k = 0;
    for (size_t i = 0; i < n; ++i)
        for (uint64_t off = 0; off < lens[i]; off += kCmpChunk) {
            CmpChunk& ch = h_tab[k++];
            ch.a = a[i] + off;
            ch.b = b[i] + off;
            ch.nbytes = (uint32_t)std::min<uint64_t>(kCmpChunk, lens[i] - off);
            ch.pair = (uint32_t)i;
        }
    HIP_TRY(c, hipMemsetAsync(d_equal, 1, n, c->stream)); // equal until a chunk says otherwise
    HIP_TRY(c, hipMemcpyAsync(d_tab, h_tab, nchunks * sizeof(CmpChunk), hipMemcpyHostToDevice, c->stream));
    EventPair* ev = next_events(c, 0);
    if (!ev) return fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
    HIP_TRY(c, hipEventRecord(ev->a, c->stream));
    hipError_t e = launch_compare(d_tab, (uint32_t)nchunks, d_equal, c->stream);
    if (e != hipSuccess) return fail(c, SNAPHASH_EDEVICE, std::string("compare launch: ") + hipGetErrorString(e));
    HIP_TRY(c, hipEventRecord(ev->b, c->stream));
    c->stats.launches++;
    c->pending = true;
    return SNAPHASH_OK;
}

struct CmpPair { size_t idx; uint64_t len, done; bool failed; };

// Reads [off, off+n) of path into dst; false on any error or short file (upstream: not equal).
bool read_exact(const char* path, uint64_t off, uint64_t n, uint8_t* dst)
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    uint64_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, dst + got, n - got, (off_t)(off + got));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        got += (uint64_t)r;
    }
    close(fd);
    return got == n;
}

int files_equal_impl(DevCtx* c, const char* const* a, const char* const* b, size_t n, uint8_t* equal)
{
    std::vector<CmpPair> todo;
    for (size_t i = 0; i < n; ++i) {
        equal[i] = 0;
        if (!a[i] || !b[i]) return fail(c, SNAPHASH_EINVAL, "NULL path");
        struct stat sa, sb; // os.Open + Stat on both; any failure or a size difference: not equal (cmp.go:31-56)
        if (stat(a[i], &sa) != 0 || stat(b[i], &sb) != 0) continue;
        if (access(a[i], R_OK) != 0 || access(b[i], R_OK) != 0) continue;
        if (sa.st_size != sb.st_size) continue;
        if (S_ISDIR(sa.st_mode) || S_ISDIR(sb.st_mode)) continue; // a read of a directory fails: not equal
        if (sa.st_size == 0) { equal[i] = 1; continue; }
        todo.push_back(CmpPair{i, (uint64_t)sa.st_size, 0, false});
        c->stats.bytes_hashed += (uint64_t)sa.st_size;
    }
    c->stats.streams = n;
    if (todo.empty()) return SNAPHASH_OK;
    uint64_t total = 0;
    for (const CmpPair& p : todo) total += p.len;
    // Small comparisons stay on the host (round 5; the planner's rule for the hash pass -- never slower than the loop it
    // replaces -- applied here): helpers.FilesAreEqual compares 4 KiB chunks on one goroutine (cmp.go:57-77), and what
    // policy.AppArmorDelta hands it is a handful of profile files of a few KiB.  Through the staging buffers that was two
    // pinned allocations (80 ms on a fresh ctx), two copies and a launch for microseconds of memcmp.  Up to 16 MiB a side
    // the pairs are read and compared by the helper threads; SNAPHASH_FLAG_GPU_ONLY keeps every byte on the kernel.
    if (!(c->owner && c->owner->gpu_only) && total <= (uint64_t)(16u << 20)) {
        const unsigned T = (unsigned)std::min<size_t>(std::max<size_t>(1, std::min<uint64_t>(todo.size(), total >> 16)), 8);
        std::atomic<size_t> next_pair{0};
        run_on_threads(T, [&](unsigned) {
            std::vector<uint8_t> ba(64u << 10), bb(64u << 10);
            auto pread_all = [](int fd, uint8_t* dst, uint64_t n, uint64_t off) {
                uint64_t got = 0;
                while (got < n) {
                    const ssize_t r = pread(fd, dst + got, n - got, (off_t)(off + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) return false; // an error, or a file that shrank since its size was taken: not equal (cmp.go:31-56)
                    got += (uint64_t)r;
                }
                return true;
            };
            for (size_t t; (t = next_pair.fetch_add(1)) < todo.size();) {
                const CmpPair& p = todo[t];
                const int fa = open(a[p.idx], O_RDONLY | O_CLOEXEC), fb = fa >= 0 ? open(b[p.idx], O_RDONLY | O_CLOEXEC) : -1;
                bool same = fa >= 0 && fb >= 0;
                for (uint64_t off = 0; off < p.len && same; off += ba.size()) {
                    const uint64_t take = std::min<uint64_t>(ba.size(), p.len - off);
                    same = pread_all(fa, ba.data(), take, off) && pread_all(fb, bb.data(), take, off) && memcmp(ba.data(), bb.data(), take) == 0;
                }
                if (fa >= 0) close(fa);
                if (fb >= 0) close(fb);
                equal[p.idx] = same ? 1 : 0;
            }
        });
        return SNAPHASH_OK;
    }
    // (the staging halves sized for the job: pinning 2 x 256 MiB for a few MiB of files cost more than comparing them)
    uint64_t slot_want = 8u << 20;
    while (slot_want < c->staging && slot_want < 2 * total) slot_want <<= 1;
    slot_want = std::min<uint64_t>(slot_want, c->staging);
    for (const Slot& sl : c->slot) slot_want = std::max<uint64_t>(slot_want, std::min<uint64_t>(sl.cap, c->staging));
    int rc = ensure_slots(c, 2, slot_want);
    if (rc) return rc;
    for (const CmpPair& p : todo) equal[p.idx] = 1; // AND-ed down batch by batch
    // Two halves of the staging buffers (A side: slot 0, B side: slot 1) alternate: the files of batch k+1
    // are read and copied while the compare kernel of batch k runs; a batch's verdicts are collected when
    // its half is needed again.
    const uint64_t H = (slot_want / 2) & ~(uint64_t)(kAlign - 1);
    struct Seg { size_t t; uint64_t at, off, n; };
    struct Half {
        std::vector<Seg> segs;
        std::vector<uint8_t> ok;
        uint8_t* h_res = nullptr; // pinned
        uint8_t* d_eq = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr, copied = nullptr;
        bool busy = false;
    } hf[2];
    auto retire = [&](Half& h) -> int {
        if (!h.busy) return SNAPHASH_OK;
        HIP_TRY(c, hipEventSynchronize(h.done));
        for (size_t i = 0; i < h.segs.size(); ++i)
            if (!h.ok[i] || !h.h_res[i]) equal[todo[h.segs[i].t].idx] = 0;
        h.busy = false;
        return SNAPHASH_OK;
    };
    auto cleanup = [&]() {
        for (Half& h : hf) {
            if (h.h_res) (void)hipHostFree(h.h_res);
            if (h.d_eq) (void)hipFree(h.d_eq);
            if (h.done) (void)hipEventDestroy(h.done);
            if (h.copied) (void)hipEventDestroy(h.copied);
        }
    };
    size_t first = 0;
    unsigned batch = 0;
    rc = SNAPHASH_OK;
    while (first < todo.size() && !rc) {
        Half& h = hf[batch & 1];
        const uint64_t base = (uint64_t)(batch & 1) * H;
        rc = retire(h);
        if (rc) break;
        h.segs.clear();
        uint64_t used = 0;
        size_t t = first;
        while (t < todo.size()) {
            const uint64_t at = (used + kAlign - 1) & ~(uint64_t)(kAlign - 1);
            if (at >= H) break;
            const uint64_t take = std::min<uint64_t>(todo[t].len - todo[t].done, (H - at) & ~(uint64_t)15);
            if (take == 0) break;
            h.segs.push_back(Seg{t, base + at, todo[t].done, take});
            used = at + take;
            todo[t].done += take;
            if (todo[t].done < todo[t].len) break; // half full mid-file: the rest goes in the next batch
            ++t;
        }
        h.ok.assign(h.segs.size(), 1);
        c->pool.parallel_for(h.segs.size(), (unsigned)std::min<size_t>(c->fill_cap, std::max<size_t>(1, h.segs.size() / 2)), [&](size_t i) {
            const Seg& g = h.segs[i];
            const CmpPair& p = todo[g.t];
            if (!read_exact(a[p.idx], g.off, g.n, c->slot[0].h_buf + g.at) ||
                !read_exact(b[p.idx], g.off, g.n, c->slot[1].h_buf + g.at))
                h.ok[i] = 0;
        });
        if (h.segs.size() > h.cap) {
            if (h.h_res) (void)hipHostFree(h.h_res);
            if (h.d_eq) (void)hipFree(h.d_eq);
            h.h_res = nullptr; h.d_eq = nullptr;
            h.cap = std::max<size_t>(h.segs.size(), 4096);
            if (hipMalloc((void**)&h.d_eq, h.cap) != hipSuccess || hipHostMalloc((void**)&h.h_res, h.cap, hipHostMallocDefault) != hipSuccess) {
                rc = fail(c, SNAPHASH_ENOMEM, "allocation of the verdict buffers failed");
                break;
            }
        }
        if (!h.done && (hipEventCreateWithFlags(&h.done, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&h.copied, hipEventDisableTiming) != hipSuccess)) {
            rc = fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
            break;
        }
        EventPair* ev = next_events(c, 1);
        if (!ev) { rc = fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed"); break; }
        if (hipEventRecord(ev->a, c->copy_stream) != hipSuccess ||
            hipMemcpyAsync(c->slot[0].d_buf + base, c->slot[0].h_buf + base, used, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
            hipMemcpyAsync(c->slot[1].d_buf + base, c->slot[1].h_buf + base, used, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
            hipEventRecord(ev->b, c->copy_stream) != hipSuccess || hipEventRecord(h.copied, c->copy_stream) != hipSuccess ||
            hipStreamWaitEvent(c->stream, h.copied, 0) != hipSuccess) {
            rc = fail(c, SNAPHASH_EDEVICE, "H2D of a comparison batch failed");
            break;
        }
        std::vector<uint64_t> va(h.segs.size()), vb(h.segs.size()), vl(h.segs.size());
        for (size_t i = 0; i < h.segs.size(); ++i) {
            va[i] = (uint64_t)(uintptr_t)(c->slot[0].d_buf + h.segs[i].at);
            vb[i] = (uint64_t)(uintptr_t)(c->slot[1].d_buf + h.segs[i].at);
            vl[i] = h.segs[i].n;
        }
        rc = launch_compare_ranges(c, va, vb, vl, h.d_eq, (int)(batch & 1));
        if (rc) break;
        if (hipMemcpyAsync(h.h_res, h.d_eq, h.segs.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipEventRecord(h.done, c->stream) != hipSuccess) {
            rc = fail(c, SNAPHASH_EDEVICE, "D2H of the verdicts failed");
            break;
        }
        h.busy = true;
        ++batch;
        first = t; // t is the first pair with bytes left
        while (first < todo.size() && todo[first].done >= todo[first].len) ++first;
    }
    for (Half& h : hf) {
        const int r2 = retire(h);
        if (!rc) rc = r2;
    }
    const int r3 = sync_ctx(c);
    if (!rc) rc = r3;
    cleanup();
    return rc;
}

} // namespace


extern "C" {

int snaphash_files_equal(snaphash_ctx* x, const char* const* a, const char* const* b, size_t n, uint8_t* equal)
try {
    if (!x || (n && (!a || !b || !equal))) return fail(x, SNAPHASH_EINVAL, "bad argument");
    TOP_ENTER(x);
    const size_t nd = x->dev.size();
    int rc = SNAPHASH_OK;
    if (nd == 1 || n < 2) {
        DevCtx* c = x->d0();
        HIP_TRY(c, hipSetDevice(c->device));
        rc = lift(x, c, files_equal_impl(c, a, b, n, equal));
    } else {
        // pairs are independent: LPT by size over the engines, one host thread each, every engine on its own link
        std::vector<uint64_t> lens(n, 0);
        for (size_t i = 0; i < n; ++i) {
            struct stat st;
            if (a[i] && stat(a[i], &st) == 0 && S_ISREG(st.st_mode)) lens[i] = (uint64_t)st.st_size;
        }
        std::vector<int32_t> shard(n);
        lpt_assign(lens.data(), n, (int)nd, shard.data());
        struct Part { std::vector<const char*> a, b; std::vector<size_t> idx; std::vector<uint8_t> eq; int rc = 0; };
        std::vector<Part> part(nd);
        for (size_t i = 0; i < n; ++i) { Part& p = part[shard[i]]; p.a.push_back(a[i]); p.b.push_back(b[i]); p.idx.push_back(i); }
        auto run = [&](size_t d) {
            Part& p = part[d];
            p.eq.assign(p.idx.size(), 0);
            DevCtx* c = x->dev[d].get();
            if (hipSetDevice(c->device) != hipSuccess) { p.rc = fail(c, SNAPHASH_EDEVICE, "hipSetDevice"); return; }
            p.rc = files_equal_impl(c, p.a.data(), p.b.data(), p.idx.size(), p.eq.data());
        };
        ThreadJoiner th;
        for (size_t d = 1; d < nd; ++d) th.spawn(run, d);
        run(0);
        th.join_all();
        for (size_t d = 0; d < nd; ++d) {
            if (part[d].rc && !rc) rc = lift(x, x->dev[d].get(), part[d].rc);
            for (size_t k = 0; k < part[d].idx.size(); ++k) equal[part[d].idx[k]] = part[d].eq[k];
        }
    }
    merge_stats(x);
    end_top(x, t_top0_);
    return rc;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_ranges_equal_device(snaphash_ctx* x, const void* d_a, const uint64_t* off_a, const void* d_b,
                                 const uint64_t* off_b, const uint64_t* lens, size_t n, void* d_equal)
try {
    if (!x || (n && (!d_a || !d_b || !off_a || !off_b || !lens || !d_equal))) return fail(x, SNAPHASH_EINVAL, "bad argument");
    if ((((uintptr_t)d_a) | ((uintptr_t)d_b)) & 15) return fail(x, SNAPHASH_EINVAL, "bases must be 16-byte aligned");
    if (x->open_batch) return fail(x, SNAPHASH_EINVAL, "a streaming batch is open on this ctx");
    DevCtx* c = x->d0();
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc) return lift(x, c, rc);
    begin_top(x);
    begin_call(c);
    if (n == 0) return SNAPHASH_OK;
    std::vector<uint64_t> va(n), vb(n), vl(n);
    for (size_t i = 0; i < n; ++i) {
        if ((off_a[i] | off_b[i]) & 15) return fail(x, SNAPHASH_EINVAL, "offsets must be 16-byte aligned");
        va[i] = (uint64_t)(uintptr_t)d_a + off_a[i];
        vb[i] = (uint64_t)(uintptr_t)d_b + off_b[i];
        vl[i] = lens[i];
        c->stats.bytes_hashed += lens[i];
    }
    c->stats.streams = n;
    return lift(x, c, launch_compare_ranges(c, va, vb, vl, (uint8_t*)d_equal));
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_dir_updated(snaphash_ctx* c, const char* dir_a, const char* dir_b, const char* pfx, char** names_out,
                         size_t* count)
try {
    if (!c || !dir_a || !dir_b || !names_out || !count) return fail(c, SNAPHASH_EINVAL, "bad argument");
    *names_out = nullptr;
    *count = 0;
    std::vector<std::string> names;
    if (DIR* d = opendir(dir_a)) { // filepath.Glob(dirA/*): every entry (leading dots too), sorted; errors ignored
        while (struct dirent* de = readdir(d))
            if (strcmp(de->d_name, ".") && strcmp(de->d_name, "..")) names.emplace_back(de->d_name);
        closedir(d);
    }
    std::sort(names.begin(), names.end());
    std::vector<std::string> pa, pb, cand;
    for (const std::string& nm : names) {
        const std::string fa = std::string(dir_a) + "/" + nm, fb = std::string(dir_b) + "/" + nm;
        struct stat st;
        if (stat(fa.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) continue; // IsDirectory(fileA): subdirectories are ignored
        if (stat(fb.c_str(), &st) != 0) continue;                         // FileExists(fileB)
        pa.push_back(fa); pb.push_back(fb); cand.push_back(nm);
    }
    std::vector<const char*> ca(pa.size()), cb(pb.size());
    for (size_t i = 0; i < pa.size(); ++i) { ca[i] = pa[i].c_str(); cb[i] = pb[i].c_str(); }
    std::vector<uint8_t> eq(pa.size());
    int rc = snaphash_files_equal(c, ca.data(), cb.data(), pa.size(), eq.data());
    if (rc) return rc;
    std::string out;
    size_t k = 0;
    for (size_t i = 0; i < cand.size(); ++i)
        if (!eq[i]) { out += pfx ? pfx : ""; out += cand[i]; out.push_back('\0'); ++k; }
    char* p = (char*)malloc(out.size() + 1);
    if (!p) return fail(c, SNAPHASH_ENOMEM, "malloc");
    memcpy(p, out.data(), out.size());
    p[out.size()] = 0;
    *names_out = p;
    *count = k;
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

// ---- host-side pieces ------------------------------------------------------------------

struct snaphash_records { std::vector<Record> v; };

int snaphash_walk(const char* build_dir, snaphash_records** out)
try {
    if (!build_dir || !out) return SNAPHASH_EINVAL;
    snaphash_records* r = new (std::nothrow) snaphash_records();
    if (!r) return SNAPHASH_ENOMEM;
    int en = 0;
    int rc = walk_tree(build_dir, r->v, &en);
    if (rc) { delete r; errno = en; return rc; }
    *out = r;
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}
size_t snaphash_records_count(const snaphash_records* r) { return r ? r->v.size() : 0; }
int snaphash_records_get(const snaphash_records* r, size_t i, snaphash_record* out)
try {
    if (!r || !out || i >= r->v.size()) return SNAPHASH_EINVAL;
    const Record& x = r->v[i];
    out->name = x.name.c_str();
    out->st_mode = x.st_mode;
    out->is_regular = x.is_regular ? 1 : 0;
    out->size = x.size;
    out->path = x.path.c_str();
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}
void snaphash_records_free(snaphash_records* r) { delete r; }

int snaphash_parse_yaml(const char* yaml, size_t yaml_len, snaphash_records** out, char archive_hex[129])
try {
    if (!yaml || !out) return SNAPHASH_EINVAL;
    ParsedHashes ph;
    int rc = parse_yaml(yaml, yaml_len, ph);
    if (rc) return rc;
    snaphash_records* r = new (std::nothrow) snaphash_records();
    if (!r) return SNAPHASH_ENOMEM;
    for (const ParsedRecord& p : ph.files) {
        Record x;
        x.name = p.name;
        x.st_mode = p.st_mode;
        x.is_regular = S_ISREG(p.st_mode);
        x.size = p.has_size ? p.size : 0;
        x.sha512_hex = p.sha512_hex;
        r->v.push_back(std::move(x));
    }
    if (archive_hex) snprintf(archive_hex, 129, "%s", ph.archive_hex.c_str());
    *out = r;
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

const char* snaphash_records_sha512_hex(const snaphash_records* r, size_t i)
{
    return (r && i < r->v.size()) ? r->v[i].sha512_hex.c_str() : "";
}

int snaphash_emit_yaml(const snaphash_records* r, const uint8_t archive_digest[64], const uint8_t* file_digests,
                       char** yaml_out, size_t* yaml_len)
try {
    if (!r || !archive_digest || !yaml_out) return SNAPHASH_EINVAL;
    if (!file_digests)
        for (const Record& rec : r->v)
            if (rec.is_regular) return SNAPHASH_EINVAL; // every regular record needs its digest
    std::string y;
    int rc = emit_yaml(r->v, archive_digest, file_digests, y);
    if (rc) return rc;
    char* p = (char*)malloc(y.size() + 1);
    if (!p) return SNAPHASH_ENOMEM;
    memcpy(p, y.data(), y.size());
    p[y.size()] = 0;
    *yaml_out = p;
    if (yaml_len) *yaml_len = y.size();
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_mode_string(uint32_t st_mode, char out[11]) { return out ? mode_string(st_mode, out) : SNAPHASH_EINVAL; }
int snaphash_mode_parse(const char* s, uint32_t* st_mode) { return (s && st_mode) ? mode_parse(s, st_mode) : SNAPHASH_EINVAL; }
int snaphash_lpt_assign(const uint64_t* lens, size_t n, int nshards, int32_t* shard_of) { return lpt_assign(lens, n, nshards, shard_of); }

int snaphash_fill_synthetic_device(snaphash_ctx* x, void* d_base, const uint64_t* offsets, const uint64_t* lens,
                                   const uint64_t* file_index, size_t n)
try {
    if (!x || (n && (!d_base || !offsets || !lens || !file_index))) return fail(x, SNAPHASH_EINVAL, "bad argument");
    if (n == 0) return SNAPHASH_OK;
    DevCtx* c = x->d0();
    HIP_TRY(c, hipSetDevice(c->device));
    uint64_t* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, 3 * n * sizeof(uint64_t)));
    uint64_t maxlen = 0;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i] & 7) { (void)hipFree(d); return fail(x, SNAPHASH_EINVAL, "offsets must be 8-byte aligned"); }
        maxlen = std::max(maxlen, lens[i]);
    }
    hipError_t e = hipMemcpyAsync(d, offsets, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, lens, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + 2 * n, file_index, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_fill_synthetic((uint8_t*)d_base, d, d + n, d + 2 * n, (uint32_t)n, maxlen, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(x, SNAPHASH_EDEVICE, std::string("fill_synthetic: ") + hipGetErrorString(e));
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

// ---- diagnostics ------------------------------------------------------------------------

const char* snaphash_strerror(int code)
{
    switch (code) {
    case SNAPHASH_OK: return "ok";
    case SNAPHASH_EINVAL: return "invalid argument";
    case SNAPHASH_ENOMEM: return "out of memory";
    case SNAPHASH_EIO: return "i/o error";
    case SNAPHASH_EDEVICE: return "no usable gfx950 device or HIP failure";
    case SNAPHASH_EMODE: return "Unknown file mode";
    case SNAPHASH_ENAME: return "file name outside the plain YAML scalar set";
    case SNAPHASH_EPARSE: return "hashes.yaml parse error";
    case SNAPHASH_EMISMATCH: return "tree does not match hashes.yaml";
    default: return "unknown error";
    }
}

const char* snaphash_last_error(const snaphash_ctx* c) { return c ? c->last_error.c_str() : g_init_error.c_str(); }

void snaphash_get_stats(const snaphash_ctx* c, snaphash_stats* out)
{
    if (c && out) *out = c->stats;
}

int snaphash_get_stats_ex(const snaphash_ctx* c, snaphash_stats_ex* out)
try {
    constexpr uint32_t kAbi4Size = (uint32_t)offsetof(snaphash_stats_ex, planned_gpu_ms);
    if (!c || !out || out->struct_size < kAbi4Size) return SNAPHASH_EINVAL;
    const uint32_t have = std::min<uint32_t>(out->struct_size, (uint32_t)sizeof(snaphash_stats_ex));
    snaphash_stats_ex v = c->ex;
    v.struct_size = have;
    v.n_devices = (uint32_t)c->dev.size();
    memcpy(out, &v, have);
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

// what an engine holds right now: pinned host bytes and HBM bytes (staging slots, job arrays, chaining values, digests,
// comparison and deflate scratch)
static void engine_footprint(const snaphash_ctx* x, size_t i, uint64_t* pinned, uint64_t* hbm)
{
    const DevCtx* d = x->dev[i].get();
    uint64_t p = 0, h = 0;
    for (const Slot& s : d->slot) {
        if (s.h_buf) p += s.cap;
        if (s.d_buf) h += s.cap + 256;
        p += s.jobs_cap * sizeof(Job);
        h += s.jobs_cap * sizeof(Job);
    }
    for (const SubSlot& q : d->sub) { // the job arrays of the batches in flight (pinned and HBM twins)
        p += q.jobs_cap * sizeof(Job);
        h += q.jobs_cap * sizeof(Job);
    }
    p += d->jobs_cap * sizeof(Job) + d->chunks_cap * sizeof(CmpChunk);
    h += d->jobs_cap * sizeof(Job) + d->chunks_cap * sizeof(CmpChunk) + d->state_cap * 64 + d->digests_cap * 64 + d->equal_cap;
    if (d->z_chunks) { // deflate scratch (targz.inc ensure_deflate)
        const uint64_t nch = d->z_chunks;
        h += 2 * nch * (uint64_t)kDeflateSlot + nch * (uint64_t)kDeflateTokWords * 4 + nch * 4 + (nch + 1) * 8;
        p += nch * 4 + (nch + 1) * 8 + 2 * nch * (uint64_t)kDeflateSlot;
    }
    if (i < x->d_gather.size() && x->d_gather[i]) h += (uint64_t)x->dev.size() * x->gather_cap * 64;
    *pinned = p;
    *hbm = h;
}

int snaphash_get_engine_info(const snaphash_ctx* c, uint32_t i, snaphash_engine_info* out)
try {
    constexpr uint32_t kAbi3Size = (uint32_t)offsetof(snaphash_engine_info, pinned_bytes);
    if (!c || !out || i >= c->dev.size() || out->struct_size < kAbi3Size) return SNAPHASH_EINVAL;
    const uint32_t have = std::min<uint32_t>(out->struct_size, (uint32_t)sizeof(snaphash_engine_info));
    const DevCtx* d = c->dev[i].get();
    snaphash_engine_info v;
    memset(&v, 0, sizeof v);
    v.struct_size = have;
    v.device = d->device;
    v.numa_node = d->numa_node;
    v.staging_node = d->staging_node;
    v.fill_threads = d->fill_cap;
    v.n_cpus = (uint32_t)d->pool.cpus().size();
    snprintf(v.pci_bus_id, sizeof v.pci_bus_id, "%s", d->pci_bus_id.c_str());
    engine_footprint(c, i, &v.pinned_bytes, &v.hbm_bytes);
    memcpy(out, &v, have);
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_get_engine_cpus(const snaphash_ctx* c, uint32_t i, int32_t* cpus, size_t cap, size_t* n_cpus)
try {
    if (!c || i >= c->dev.size() || !n_cpus) return SNAPHASH_EINVAL;
    const std::vector<int>& v = c->dev[i]->pool.cpus();
    *n_cpus = v.size();
    for (size_t k = 0; cpus && k < v.size() && k < cap; ++k) cpus[k] = v[k];
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

int snaphash_numa_slice(const char* sysfs_root, int32_t node, uint32_t pos, uint32_t m, int32_t* cpus, size_t cap, size_t* n_cpus)
try {
    if (!sysfs_root || !n_cpus) return SNAPHASH_EINVAL;
    const std::vector<int> v = slice_cpus(numa_cpus_of_node(sysfs_root, node), pos, m);
    *n_cpus = v.size();
    for (size_t k = 0; cpus && k < v.size() && k < cap; ++k) cpus[k] = v[k];
    return SNAPHASH_OK;
} catch (...) {
    return SNAPHASH_ENOMEM;
}

int snaphash_numa_probe(const char* sysfs_root, const char* pci_bus_id, int32_t* node, int32_t* cpus, size_t cap, size_t* n_cpus)
try {
    if (!sysfs_root || !pci_bus_id || !node) return SNAPHASH_EINVAL;
    *node = numa_node_of_pci(sysfs_root, pci_bus_id);
    const std::vector<int> v = numa_cpus_of_node(sysfs_root, *node);
    if (n_cpus) *n_cpus = v.size();
    for (size_t k = 0; cpus && k < v.size() && k < cap; ++k) cpus[k] = v[k];
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

int snaphash_get_device_stats(const snaphash_ctx* c, uint32_t i, int32_t* device, snaphash_stats* out)
try {
    if (!c || i >= c->dev.size()) return SNAPHASH_EINVAL;
    if (device) *device = c->dev[i]->device;
    if (out) *out = c->dev[i]->stats;
    return SNAPHASH_OK;
} catch (...) { // allocation or thread-creation failure: no C++ exception crosses the C boundary
    return SNAPHASH_ENOMEM;
}

} // extern "C"
