// snaphash_api.cpp -- the C ABI of libsnaphash.so (include/snaphash.h): context,
// HBM/pinned staging, the chunked double-buffered streaming engine that feeds
// the multi-buffer SHA-512 kernels, and the writeHashes / Verify passes built
// on it.
//
// Reference behaviour mirrored here (paths relative to the upstream tree):
//   helpers/helpers.go:187-201  Sha512sum: whole-file digest, any read error fails
//   snappy/build.go:216-270     writeHashes: archive digest first, then the walk;
//                               the first error aborts, nothing is written
// Hashing happens on the GPU only; there is no host fallback in this file.
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "hostpass.h"
#include "sha512_core.h"
#include "sha512_kernels.h"

using namespace snaphash;

namespace {

constexpr uint64_t kDefaultStaging = 256ull << 20;
constexpr uint64_t kAlign = 256;          // placement of a segment inside a staging buffer
constexpr uint32_t kTargetStreams = 4096; // streams per batch the engine aims for (keeps the kernel ahead of PCIe)

struct EventPair { hipEvent_t a = nullptr, b = nullptr; int kind = 0; }; // kind 0 kernel, 1 h2d

struct Slot {
    uint8_t* h_buf = nullptr; // pinned host
    uint8_t* d_buf = nullptr; // HBM
    Job* h_jobs = nullptr;    // pinned
    Job* d_jobs = nullptr;
    size_t jobs_cap = 0;
    hipEvent_t done = nullptr;   // kernel of the batch staged in this slot has finished
    hipEvent_t copied = nullptr; // H2D of this slot's data + jobs has finished
    bool busy = false;
};

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct snaphash_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr; // H2D of batch k+1 overlaps the kernel of batch k
    uint64_t staging = kDefaultStaging;
    uint32_t kernel_pref = SNAPHASH_KERNEL_AUTO;

    Slot slot[2];
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

    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    bool pending = false;

    snaphash_stats stats{};
    double t_call0 = 0;
    std::string last_error;
};

namespace {

int fail(snaphash_ctx* c, int code, const std::string& msg)
{
    if (c) c->last_error = msg;
    return code;
}
#define HIP_TRY(c, expr)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(c, (e_ == hipErrorOutOfMemory) ? SNAPHASH_ENOMEM : SNAPHASH_EDEVICE,         \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                          \
    } while (0)

EventPair* next_events(snaphash_ctx* c, int kind)
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

void collect_events(snaphash_ctx* c)
{
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) == hipSuccess) {
            if (c->ev_pool[i].kind == 0) c->stats.kernel_ms += ms;
            else c->stats.h2d_ms += ms;
        }
    }
    c->ev_used = 0;
}

int ensure_state(snaphash_ctx* c, size_t n, bool want_digests)
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

int ensure_jobs(snaphash_ctx* c, Job** h, Job** d, size_t* cap, size_t n)
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

int ensure_slots(snaphash_ctx* c)
{
    for (Slot& s : c->slot) {
        if (!s.h_buf) HIP_TRY(c, hipHostMalloc((void**)&s.h_buf, c->staging, hipHostMallocDefault));
        if (!s.d_buf) HIP_TRY(c, hipMalloc((void**)&s.d_buf, c->staging));
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

hipError_t launch_kernel(uint32_t k, const Job* d_jobs, size_t n, snaphash_ctx* c, uint8_t* d_digests)
{
    if (k == SNAPHASH_KERNEL_PAIR) return launch_pair(d_jobs, (uint32_t)n, c->d_state, d_digests, c->stream);
    if (k == SNAPHASH_KERNEL_SPLIT) return launch_split(d_jobs, (uint32_t)n, c->d_state, d_digests, c->stream);
    return launch_wide(d_jobs, (uint32_t)n, c->d_state, d_digests, c->stream);
}

// h_jobs is sorted longest first.  Returns how many leading jobs go to the
// few-long-streams kernel (*k_head) and which kernel takes the rest (*k_tail).
size_t plan_kernels(const snaphash_ctx* c, const Job* h_jobs, size_t n, uint32_t* k_head, uint32_t* k_tail)
{
    *k_tail = SNAPHASH_KERNEL_WIDE;
    if (c->kernel_pref == SNAPHASH_KERNEL_WIDE || c->kernel_pref == SNAPHASH_KERNEL_SPLIT ||
        c->kernel_pref == SNAPHASH_KERNEL_PAIR) {
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
int launch_jobs(snaphash_ctx* c, Job* h_jobs, Job* d_jobs, size_t n, uint8_t* d_digests, hipEvent_t copied = nullptr)
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
    hipError_t e = launch_kernel(k_head, d_jobs, head, c, d_digests);
    if (e == hipSuccess && head < n) e = launch_kernel(k_tail, d_jobs + head, n - head, c, d_digests);
    if (e != hipSuccess) return fail(c, SNAPHASH_EDEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
    HIP_TRY(c, hipEventRecord(ev->b, c->stream));
    c->stats.launches += (head < n) ? 2 : 1;
    c->stats.kernel_used = k_head;
    c->pending = true;
    return SNAPHASH_OK;
}

int sync_ctx(snaphash_ctx* c)
{
    if (!c->pending && c->ev_used == 0) return SNAPHASH_OK;
    if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    c->pending = false;
    if (c->t_call0 > 0) { c->stats.wall_ms = now_ms() - c->t_call0; c->t_call0 = 0; }
    return SNAPHASH_OK;
}

void begin_call(snaphash_ctx* c)
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
};

struct ReadOp { uint32_t src; uint64_t off; uint64_t n; uint8_t* dst; };

void run_reads(const std::vector<Source>& src, const std::vector<ReadOp>& ops, std::atomic<int>& first_err,
               std::atomic<int64_t>& first_err_src)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::min<size_t>(std::min(16u, hw), std::max<size_t>(1, ops.size() / 4));
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= ops.size() || first_err.load()) return;
            const ReadOp& op = ops[i];
            const Source& s = src[op.src];
            if (s.mem) { memcpy(op.dst, s.mem + op.off, op.n); continue; }
            int err = 0;
            int fd = open(s.path, O_RDONLY | O_CLOEXEC);
            if (fd < 0) err = errno;
            uint64_t got = 0;
            while (!err && got < op.n) {
                ssize_t r = pread(fd, op.dst + got, op.n - got, (off_t)(op.off + got));
                if (r < 0) { if (errno == EINTR) continue; err = errno; }
                else if (r == 0) err = EIO; // file shrank underneath us
                else got += (uint64_t)r;
            }
            if (fd >= 0) close(fd);
            if (err) {
                int z = 0;
                if (first_err.compare_exchange_strong(z, err)) first_err_src.store(op.src);
                return;
            }
        }
    };
    if (T <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

int hash_sources(snaphash_ctx* c, const std::vector<Source>& src, uint8_t* digests, int32_t* status)
{
    const size_t n = src.size();
    if (n == 0) return SNAPHASH_OK;
    int rc = ensure_slots(c);
    if (rc) return rc;
    rc = ensure_state(c, n, true);
    if (rc) return rc;

    std::vector<uint64_t> done(n, 0);
    std::vector<uint32_t> active(n);
    for (size_t i = 0; i < n; ++i) active[i] = (uint32_t)i;
    std::atomic<int> first_err{0};
    std::atomic<int64_t> first_err_src{-1};
    std::vector<ReadOp> ops;
    const uint64_t S = c->staging;
    unsigned batch = 0;

    while (!active.empty()) {
        Slot& sl = c->slot[batch & 1];
        if (sl.busy) { HIP_TRY(c, hipEventSynchronize(sl.done)); sl.busy = false; }
        const uint64_t target = std::min<uint64_t>(active.size(), kTargetStreams);
        uint64_t quota = (S / target) & ~(uint64_t)(kAlign - 1);
        if (quota < kAlign) quota = kAlign;
        rc = ensure_jobs(c, &sl.h_jobs, &sl.d_jobs, &sl.jobs_cap, active.size());
        if (rc) return rc;

        ops.clear();
        size_t nj = 0;
        uint64_t used = 0;
        std::vector<uint32_t> still;
        still.reserve(active.size());
        bool full = false;
        for (uint32_t id : active) {
            if (full) { still.push_back(id); continue; }
            const uint64_t rem = src[id].len - done[id];
            const uint64_t take = rem <= quota ? rem : quota; // quota is a multiple of 128
            const uint64_t at = (used + kAlign - 1) & ~(uint64_t)(kAlign - 1);
            if (at + take > S) { full = true; still.push_back(id); continue; }
            Job j;
            j.data = (uint64_t)(uintptr_t)(sl.d_buf + at);
            j.nbytes = take;
            j.total_prev = done[id];
            j.idx = id;
            j.flags = (done[id] == 0 ? kJobFirst : 0u) | (take == rem ? kJobFinal : 0u);
            sl.h_jobs[nj++] = j;
            if (take) ops.push_back(ReadOp{id, done[id], take, sl.h_buf + at});
            used = at + take;
            done[id] += take;
            c->stats.blocks += padded_blocks(take, take == rem);
            if (take != rem) still.push_back(id);
        }
        active.swap(still);

        run_reads(src, ops, first_err, first_err_src);
        if (first_err.load()) break;

        if (used) { // copy stream: the slot's previous kernel was already waited for above
            EventPair* ev = next_events(c, 1);
            if (!ev) return fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
            HIP_TRY(c, hipEventRecord(ev->a, c->copy_stream));
            HIP_TRY(c, hipMemcpyAsync(sl.d_buf, sl.h_buf, used, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(c, hipEventRecord(ev->b, c->copy_stream));
        }
        rc = launch_jobs(c, sl.h_jobs, sl.d_jobs, nj, c->d_digests, sl.copied);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(sl.done, c->stream));
        sl.busy = true;
        ++batch;
    }

    rc = sync_ctx(c);
    c->slot[0].busy = c->slot[1].busy = false;
    if (rc) return rc;
    if (first_err.load()) {
        const int64_t s = first_err_src.load();
        if (status) {
            for (size_t i = 0; i < n; ++i) status[i] = 0;
            if (s >= 0) status[s] = first_err.load();
        }
        return fail(c, SNAPHASH_EIO,
                    std::string(s >= 0 && src[s].path ? src[s].path : "<buffer>") + ": " + strerror(first_err.load()));
    }
    HIP_TRY(c, hipMemcpy(digests, c->d_digests, n * 64, hipMemcpyDeviceToHost));
    if (status) for (size_t i = 0; i < n; ++i) status[i] = 0;
    for (size_t i = 0; i < n; ++i) c->stats.bytes_hashed += src[i].len;
    c->stats.streams = n;
    return SNAPHASH_OK;
}

int hash_paths(snaphash_ctx* c, const char* const* paths, size_t n, uint8_t* digests, int32_t* status)
{
    std::vector<Source> src(n);
    for (size_t i = 0; i < n; ++i) {
        if (!paths[i]) return fail(c, SNAPHASH_EINVAL, "NULL path");
        struct stat st;
        int err = 0;
        // os.Open follows symlinks (helpers.go:189); a directory opens but its read fails with EISDIR
        if (stat(paths[i], &st) != 0) err = errno;
        else if (S_ISDIR(st.st_mode)) err = EISDIR;
        else if (access(paths[i], R_OK) != 0) err = errno;
        if (err) {
            if (status) { for (size_t k = 0; k < n; ++k) status[k] = 0; status[i] = err; }
            return fail(c, SNAPHASH_EIO, std::string(paths[i]) + ": " + strerror(err));
        }
        src[i].path = paths[i];
        src[i].len = (uint64_t)st.st_size;
    }
    return hash_sources(c, src, digests, status);
}

} // namespace

// ================================== C ABI ==========================================

extern "C" {

int snaphash_abi_version(void) { return SNAPHASH_ABI_VERSION; }

static thread_local std::string g_init_error; // why the last snaphash_init on this thread failed

static int init_fail(int code, const std::string& what, hipError_t e = hipSuccess)
{
    g_init_error = what + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
    return code;
}

int snaphash_init(const snaphash_config* cfg, snaphash_ctx** out)
{
    if (!out) return SNAPHASH_EINVAL;
    *out = nullptr;
    g_init_error.clear();
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return init_fail(SNAPHASH_EDEVICE, "hipGetDeviceCount found no device", e);
    int dev = -1;
    if (cfg && cfg->struct_size >= sizeof(snaphash_config)) dev = cfg->device;
    if (dev < 0 && (e = hipGetDevice(&dev)) != hipSuccess) return init_fail(SNAPHASH_EDEVICE, "hipGetDevice", e);
    if (dev >= ndev) return init_fail(SNAPHASH_EINVAL, "device ordinal out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return init_fail(SNAPHASH_EDEVICE, "hipGetDeviceProperties", e);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) // the code object is gfx950-only
        return init_fail(SNAPHASH_EDEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    if ((e = hipSetDevice(dev)) != hipSuccess) return init_fail(SNAPHASH_EDEVICE, "hipSetDevice", e);
    snaphash_ctx* c = new (std::nothrow) snaphash_ctx();
    if (!c) return SNAPHASH_ENOMEM;
    c->device = dev;
    if (cfg && cfg->struct_size >= sizeof(snaphash_config)) {
        if (cfg->staging_bytes) c->staging = (cfg->staging_bytes + kAlign - 1) & ~(uint64_t)(kAlign - 1);
        c->kernel_pref = cfg->kernel;
        if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
    }
    if (c->staging < (1u << 16)) c->staging = 1u << 16;
    if (!c->stream) {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            delete c;
            return init_fail(SNAPHASH_EDEVICE, "hipStreamCreateWithFlags", e);
        }
        c->own_stream = true;
    }
    *out = c;
    return SNAPHASH_OK;
}

void snaphash_destroy(snaphash_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (Slot& s : c->slot) {
        if (s.h_buf) (void)hipHostFree(s.h_buf);
        if (s.d_buf) (void)hipFree(s.d_buf);
        if (s.h_jobs) (void)hipHostFree(s.h_jobs);
        if (s.d_jobs) (void)hipFree(s.d_jobs);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.copied) (void)hipEventDestroy(s.copied);
    }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->h_chunks) (void)hipHostFree(c->h_chunks);
    if (c->d_chunks) (void)hipFree(c->d_chunks);
    if (c->d_equal) (void)hipFree(c->d_equal);
    if (c->h_jobs) (void)hipHostFree(c->h_jobs);
    if (c->d_jobs) (void)hipFree(c->d_jobs);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_digests) (void)hipFree(c->d_digests);
    for (EventPair& p : c->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int snaphash_sha512_files(snaphash_ctx* c, const char* const* paths, size_t n, uint8_t* digests, int32_t* status)
{
    if (!c || (n && (!paths || !digests))) return fail(c, SNAPHASH_EINVAL, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc) return rc;
    begin_call(c);
    rc = hash_paths(c, paths, n, digests, status);
    if (c->t_call0 > 0) { c->stats.wall_ms = now_ms() - c->t_call0; c->t_call0 = 0; }
    return rc;
}

int snaphash_sha512_buffers(snaphash_ctx* c, const void* const* bufs, const uint64_t* lens, size_t n, uint8_t* digests)
{
    if (!c || (n && (!bufs || !lens || !digests))) return fail(c, SNAPHASH_EINVAL, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc) return rc;
    begin_call(c);
    std::vector<Source> src(n);
    for (size_t i = 0; i < n; ++i) {
        if (!bufs[i] && lens[i]) return fail(c, SNAPHASH_EINVAL, "NULL buffer with non-zero length");
        src[i].mem = bufs[i] ? (const uint8_t*)bufs[i] : (const uint8_t*)"";
        src[i].len = lens[i];
    }
    rc = hash_sources(c, src, digests, nullptr);
    if (c->t_call0 > 0) { c->stats.wall_ms = now_ms() - c->t_call0; c->t_call0 = 0; }
    return rc;
}

int snaphash_sha512_device(snaphash_ctx* c, const void* d_base, const uint64_t* offsets, const uint64_t* lens,
                           size_t n, void* d_digests)
{
    if (!c || (n && (!d_base || !offsets || !lens || !d_digests))) return fail(c, SNAPHASH_EINVAL, "bad argument");
    if (((uintptr_t)d_base & 15) != 0) return fail(c, SNAPHASH_EINVAL, "d_base must be 16-byte aligned");
    if (n > 0xffffffffull) return fail(c, SNAPHASH_EINVAL, "too many streams");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c); // one call in flight per ctx: the pinned job array is reused
    if (rc) return rc;
    begin_call(c);
    if (n == 0) return SNAPHASH_OK;
    rc = ensure_state(c, n, false);
    if (rc) return rc;
    rc = ensure_jobs(c, &c->h_jobs, &c->d_jobs, &c->jobs_cap, n);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i] & 15) return fail(c, SNAPHASH_EINVAL, "offsets must be 16-byte aligned");
        if (lens[i] >> 35) return fail(c, SNAPHASH_EINVAL, "a resident stream is limited to 32 GiB per call (32-bit block counters)");
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
    return launch_jobs(c, c->h_jobs, c->d_jobs, n, (uint8_t*)d_digests);
}

int snaphash_sync(snaphash_ctx* c)
{
    if (!c) return SNAPHASH_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    return sync_ctx(c);
}

// ---- the pass -----------------------------------------------------------------------

static int tree_impl(snaphash_ctx* c, const char* build_dir, const char* data_tar, std::string& yaml)
{
    // build.go:222 hashes the archive first; a missing archive fails before the walk
    struct stat st;
    if (stat(data_tar, &st) != 0) return fail(c, SNAPHASH_EIO, std::string(data_tar) + ": " + strerror(errno));
    std::vector<Record> recs;
    int en = 0;
    int rc = walk_tree(build_dir, recs, &en);
    if (rc) return fail(c, rc, rc == SNAPHASH_EIO ? std::string(build_dir) + ": " + strerror(en) : "Unknown file mode");
    for (const Record& r : recs)
        if (!plain_safe_name(r.name)) return fail(c, SNAPHASH_ENAME, "name needs YAML quoting: " + r.name);
    std::vector<const char*> paths;
    paths.push_back(data_tar); // element 0 = the archive, as in the Go batch shape
    for (const Record& r : recs)
        if (r.is_regular) paths.push_back(r.path.c_str());
    std::vector<uint8_t> dig(paths.size() * 64);
    rc = hash_paths(c, paths.data(), paths.size(), dig.data(), nullptr);
    if (rc) return rc;
    // io.Copy reads to EOF, info.Size() comes from lstat: a file that grew or shrank
    // between the two is an error here rather than a silently inconsistent record.
    return emit_yaml(recs, dig.data(), dig.data() + 64, yaml);
}

int snaphash_tree(snaphash_ctx* c, const char* build_dir, const char* data_tar, char** yaml_out, size_t* yaml_len)
{
    if (!c || !build_dir || !data_tar || !yaml_out) return fail(c, SNAPHASH_EINVAL, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc) return rc;
    begin_call(c);
    std::string y;
    rc = tree_impl(c, build_dir, data_tar, y);
    if (c->t_call0 > 0) { c->stats.wall_ms = now_ms() - c->t_call0; c->t_call0 = 0; }
    if (rc) return rc;
    char* p = (char*)malloc(y.size() + 1);
    if (!p) return fail(c, SNAPHASH_ENOMEM, "malloc");
    memcpy(p, y.data(), y.size());
    p[y.size()] = 0;
    *yaml_out = p;
    if (yaml_len) *yaml_len = y.size();
    return SNAPHASH_OK;
}

int snaphash_write_hashes(snaphash_ctx* c, const char* build_dir, const char* data_tar)
{
    if (!c || !build_dir || !data_tar) return fail(c, SNAPHASH_EINVAL, "bad argument");
    std::string dir = std::string(build_dir) + "/DEBIAN";
    (void)mkdir(dir.c_str(), 0755); // os.MkdirAll(debianDir, 0755), error ignored (build.go:219)
    char* y = nullptr;
    size_t n = 0;
    int rc = snaphash_tree(c, build_dir, data_tar, &y, &n);
    if (rc) return rc; // nothing is written on error (build.go:260-267)
    const std::string path = dir + "/hashes.yaml";
    int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644); // ioutil.WriteFile(..., 0644)
    if (fd < 0) { free(y); return fail(c, SNAPHASH_EIO, path + ": " + strerror(errno)); }
    size_t off = 0;
    while (off < n) {
        ssize_t w = write(fd, y + off, n - off);
        if (w < 0) { if (errno == EINTR) continue; int e = errno; close(fd); free(y); return fail(c, SNAPHASH_EIO, path + ": " + strerror(e)); }
        off += (size_t)w;
    }
    close(fd);
    free(y);
    return SNAPHASH_OK;
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

int snaphash_verify(snaphash_ctx* c, const char* inst_dir, const char* data_tar, const char* yaml, size_t yaml_len,
                    snaphash_mismatch* first)
{
    if (!c || !inst_dir || !yaml) return fail(c, SNAPHASH_EINVAL, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc) return rc;
    begin_call(c);
    ParsedHashes ph;
    rc = parse_yaml(yaml, yaml_len, ph);
    if (rc) return fail(c, rc, "hashes.yaml: parse error");
    std::vector<Record> recs;
    int en = 0;
    rc = walk_tree(inst_dir, recs, &en);
    if (rc) return fail(c, rc, rc == SNAPHASH_EIO ? std::string(inst_dir) + ": " + strerror(en) : "Unknown file mode");

    // Both lists are in walk order (per-directory byte-wise pre-order), so a
    // merge-style scan finds the first name present on one side only.
    size_t i = 0, j = 0;
    while (i < ph.files.size() && j < recs.size()) {
        if (ph.files[i].name == recs[j].name) { ++i; ++j; continue; }
        // decide which side is "extra": look the yaml name up on disk
        bool on_disk = false;
        for (size_t k = j; k < recs.size(); ++k)
            if (recs[k].name == ph.files[i].name) { on_disk = true; break; }
        return on_disk ? mismatch(c, first, 2, recs[j].name) : mismatch(c, first, 1, ph.files[i].name);
    }
    if (i < ph.files.size()) return mismatch(c, first, 1, ph.files[i].name);
    if (j < recs.size()) return mismatch(c, first, 2, recs[j].name);

    for (size_t k = 0; k < recs.size(); ++k) {
        const ParsedRecord& p = ph.files[k];
        const Record& r = recs[k];
        char a[11], b[11];
        if (mode_string(p.st_mode, a) || mode_string(r.st_mode, b) || memcmp(a, b, 10) != 0) return mismatch(c, first, 5, r.name);
        if (r.is_regular) {
            if (!p.has_size || p.size != r.size) return mismatch(c, first, 3, r.name);
        } else if (p.has_size || !p.sha512_hex.empty()) {
            return mismatch(c, first, 3, r.name);
        }
    }
    std::vector<const char*> paths;
    std::vector<size_t> owner;
    const bool check_archive = data_tar && ph.has_archive;
    if (check_archive) { paths.push_back(data_tar); owner.push_back((size_t)-1); }
    for (size_t k = 0; k < recs.size(); ++k)
        if (recs[k].is_regular) { paths.push_back(recs[k].path.c_str()); owner.push_back(k); }
    std::vector<uint8_t> dig(paths.size() * 64);
    rc = hash_paths(c, paths.data(), paths.size(), dig.data(), nullptr);
    if (c->t_call0 > 0) { c->stats.wall_ms = now_ms() - c->t_call0; c->t_call0 = 0; }
    if (rc) return rc;
    for (size_t q = 0; q < paths.size(); ++q) {
        if (owner[q] == (size_t)-1) {
            if (!digest_matches_hex(dig.data() + 64 * q, ph.archive_hex)) return mismatch(c, first, 6, "archive-sha512");
        } else if (!digest_matches_hex(dig.data() + 64 * q, ph.files[owner[q]].sha512_hex)) {
            return mismatch(c, first, 4, recs[owner[q]].name);
        }
    }
    return SNAPHASH_OK;
}

void snaphash_free(void* p) { free(p); }

// ---- helpers.FilesAreEqual / DirUpdated (row f4) ----------------------------------------------

namespace {

int ensure_chunks(snaphash_ctx* c, size_t n)
{
    if (n <= c->chunks_cap) return SNAPHASH_OK;
    const size_t want = std::max<size_t>(n, 4096);
    if (c->h_chunks) (void)hipHostFree(c->h_chunks);
    if (c->d_chunks) (void)hipFree(c->d_chunks);
    c->h_chunks = nullptr; c->d_chunks = nullptr; c->chunks_cap = 0;
    HIP_TRY(c, hipHostMalloc((void**)&c->h_chunks, want * sizeof(CmpChunk), hipHostMallocDefault));
    HIP_TRY(c, hipMalloc((void**)&c->d_chunks, want * sizeof(CmpChunk)));
    c->chunks_cap = want;
    return SNAPHASH_OK;
}

// Ranges (device addresses) -> chunk table -> kernel.  d_equal must hold one byte per pair.
int launch_compare_ranges(snaphash_ctx* c, const std::vector<uint64_t>& a, const std::vector<uint64_t>& b,
                          const std::vector<uint64_t>& lens, uint8_t* d_equal)
{
    const size_t n = lens.size();
    size_t nchunks = 0;
    for (size_t i = 0; i < n; ++i) nchunks += (size_t)((lens[i] + kCmpChunk - 1) / kCmpChunk);
    if (nchunks > 0xffffffffull) return fail(c, SNAPHASH_EINVAL, "too many comparison chunks");
    int rc = ensure_chunks(c, nchunks);
    if (rc) return rc;
    size_t k = 0;
    for (size_t i = 0; i < n; ++i)
        for (uint64_t off = 0; off < lens[i]; off += kCmpChunk) {
            CmpChunk& ch = c->h_chunks[k++];
            ch.a = a[i] + off;
            ch.b = b[i] + off;
            ch.nbytes = (uint32_t)std::min<uint64_t>(kCmpChunk, lens[i] - off);
            ch.pair = (uint32_t)i;
        }
    HIP_TRY(c, hipMemsetAsync(d_equal, 1, n, c->stream)); // equal until a chunk says otherwise
    HIP_TRY(c, hipMemcpyAsync(c->d_chunks, c->h_chunks, nchunks * sizeof(CmpChunk), hipMemcpyHostToDevice, c->stream));
    EventPair* ev = next_events(c, 0);
    if (!ev) return fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
    HIP_TRY(c, hipEventRecord(ev->a, c->stream));
    hipError_t e = launch_compare(c->d_chunks, (uint32_t)nchunks, d_equal, c->stream);
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

int files_equal_impl(snaphash_ctx* c, const char* const* a, const char* const* b, size_t n, uint8_t* equal)
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
    int rc = ensure_slots(c);
    if (rc) return rc;
    for (const CmpPair& p : todo) equal[p.idx] = 1; // AND-ed down batch by batch
    const uint64_t S = c->staging;
    size_t first = 0;
    while (first < todo.size()) {
        // pack segments of consecutive pairs into the two staging buffers (A side: slot 0, B side: slot 1)
        struct Seg { size_t t; uint64_t at, off, n; };
        std::vector<Seg> segs;
        uint64_t used = 0;
        size_t t = first;
        while (t < todo.size()) {
            const uint64_t at = (used + kAlign - 1) & ~(uint64_t)(kAlign - 1);
            if (at >= S) break;
            const uint64_t take = std::min<uint64_t>(todo[t].len - todo[t].done, (S - at) & ~(uint64_t)15);
            if (take == 0) break;
            segs.push_back(Seg{t, at, todo[t].done, take});
            used = at + take;
            todo[t].done += take;
            if (todo[t].done < todo[t].len) break; // buffer full mid-file: the rest goes in the next batch
            ++t;
        }
        std::vector<uint8_t> ok(segs.size(), 1);
        {
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= segs.size()) return;
                    const Seg& g = segs[i];
                    const CmpPair& p = todo[g.t];
                    if (!read_exact(a[p.idx], g.off, g.n, c->slot[0].h_buf + g.at) ||
                        !read_exact(b[p.idx], g.off, g.n, c->slot[1].h_buf + g.at))
                        ok[i] = 0;
                }
            };
            const unsigned T = (unsigned)std::min<size_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())),
                                                          std::max<size_t>(1, segs.size() / 2));
            std::vector<std::thread> th;
            for (unsigned k = 1; k < T; ++k) th.emplace_back(worker);
            worker();
            for (auto& x : th) x.join();
        }
        if (segs.size() > c->equal_cap) {
            if (c->d_equal) (void)hipFree(c->d_equal);
            c->d_equal = nullptr; c->equal_cap = 0;
            HIP_TRY(c, hipMalloc((void**)&c->d_equal, std::max<size_t>(segs.size(), 4096)));
            c->equal_cap = std::max<size_t>(segs.size(), 4096);
        }
        EventPair* ev = next_events(c, 1);
        if (!ev) return fail(c, SNAPHASH_EDEVICE, "hipEventCreate failed");
        HIP_TRY(c, hipEventRecord(ev->a, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->slot[0].d_buf, c->slot[0].h_buf, used, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->slot[1].d_buf, c->slot[1].h_buf, used, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(ev->b, c->stream));
        std::vector<uint64_t> va(segs.size()), vb(segs.size()), vl(segs.size());
        for (size_t i = 0; i < segs.size(); ++i) {
            va[i] = (uint64_t)(uintptr_t)(c->slot[0].d_buf + segs[i].at);
            vb[i] = (uint64_t)(uintptr_t)(c->slot[1].d_buf + segs[i].at);
            vl[i] = segs[i].n;
        }
        rc = launch_compare_ranges(c, va, vb, vl, c->d_equal);
        if (rc) return rc;
        std::vector<uint8_t> res(segs.size());
        HIP_TRY(c, hipMemcpyAsync(res.data(), c->d_equal, segs.size(), hipMemcpyDeviceToHost, c->stream));
        rc = sync_ctx(c);
        if (rc) return rc;
        for (size_t i = 0; i < segs.size(); ++i)
            if (!ok[i] || !res[i]) equal[todo[segs[i].t].idx] = 0;
        first = (t < todo.size() && todo[t].done < todo[t].len) ? t : t; // t is the first pair with bytes left
        while (first < todo.size() && todo[first].done >= todo[first].len) ++first;
    }
    return SNAPHASH_OK;
}

} // namespace

int snaphash_files_equal(snaphash_ctx* c, const char* const* a, const char* const* b, size_t n, uint8_t* equal)
{
    if (!c || (n && (!a || !b || !equal))) return fail(c, SNAPHASH_EINVAL, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc) return rc;
    begin_call(c);
    rc = files_equal_impl(c, a, b, n, equal);
    if (c->t_call0 > 0) { c->stats.wall_ms = now_ms() - c->t_call0; c->t_call0 = 0; }
    return rc;
}

int snaphash_ranges_equal_device(snaphash_ctx* c, const void* d_a, const uint64_t* off_a, const void* d_b,
                                 const uint64_t* off_b, const uint64_t* lens, size_t n, void* d_equal)
{
    if (!c || (n && (!d_a || !d_b || !off_a || !off_b || !lens || !d_equal))) return fail(c, SNAPHASH_EINVAL, "bad argument");
    if ((((uintptr_t)d_a) | ((uintptr_t)d_b)) & 15) return fail(c, SNAPHASH_EINVAL, "bases must be 16-byte aligned");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc) return rc;
    begin_call(c);
    if (n == 0) return SNAPHASH_OK;
    std::vector<uint64_t> va(n), vb(n), vl(n);
    for (size_t i = 0; i < n; ++i) {
        if ((off_a[i] | off_b[i]) & 15) return fail(c, SNAPHASH_EINVAL, "offsets must be 16-byte aligned");
        va[i] = (uint64_t)(uintptr_t)d_a + off_a[i];
        vb[i] = (uint64_t)(uintptr_t)d_b + off_b[i];
        vl[i] = lens[i];
        c->stats.bytes_hashed += lens[i];
    }
    c->stats.streams = n;
    return launch_compare_ranges(c, va, vb, vl, (uint8_t*)d_equal);
}

int snaphash_dir_updated(snaphash_ctx* c, const char* dir_a, const char* dir_b, const char* pfx, char** names_out,
                         size_t* count)
{
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
}

// ---- host-side pieces ------------------------------------------------------------------

struct snaphash_records { std::vector<Record> v; };

int snaphash_walk(const char* build_dir, snaphash_records** out)
{
    if (!build_dir || !out) return SNAPHASH_EINVAL;
    snaphash_records* r = new (std::nothrow) snaphash_records();
    if (!r) return SNAPHASH_ENOMEM;
    int en = 0;
    int rc = walk_tree(build_dir, r->v, &en);
    if (rc) { delete r; errno = en; return rc; }
    *out = r;
    return SNAPHASH_OK;
}
size_t snaphash_records_count(const snaphash_records* r) { return r ? r->v.size() : 0; }
int snaphash_records_get(const snaphash_records* r, size_t i, snaphash_record* out)
{
    if (!r || !out || i >= r->v.size()) return SNAPHASH_EINVAL;
    const Record& x = r->v[i];
    out->name = x.name.c_str();
    out->st_mode = x.st_mode;
    out->is_regular = x.is_regular ? 1 : 0;
    out->size = x.size;
    out->path = x.path.c_str();
    return SNAPHASH_OK;
}
void snaphash_records_free(snaphash_records* r) { delete r; }

int snaphash_parse_yaml(const char* yaml, size_t yaml_len, snaphash_records** out, char archive_hex[129])
{
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
}

const char* snaphash_records_sha512_hex(const snaphash_records* r, size_t i)
{
    return (r && i < r->v.size()) ? r->v[i].sha512_hex.c_str() : "";
}

int snaphash_emit_yaml(const snaphash_records* r, const uint8_t archive_digest[64], const uint8_t* file_digests,
                       char** yaml_out, size_t* yaml_len)
{
    if (!r || !archive_digest || !yaml_out) return SNAPHASH_EINVAL;
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
}

int snaphash_mode_string(uint32_t st_mode, char out[11]) { return out ? mode_string(st_mode, out) : SNAPHASH_EINVAL; }
int snaphash_mode_parse(const char* s, uint32_t* st_mode) { return (s && st_mode) ? mode_parse(s, st_mode) : SNAPHASH_EINVAL; }
int snaphash_lpt_assign(const uint64_t* lens, size_t n, int nshards, int32_t* shard_of) { return lpt_assign(lens, n, nshards, shard_of); }

int snaphash_fill_synthetic_device(snaphash_ctx* c, void* d_base, const uint64_t* offsets, const uint64_t* lens,
                                   const uint64_t* file_index, size_t n)
{
    if (!c || (n && (!d_base || !offsets || !lens || !file_index))) return fail(c, SNAPHASH_EINVAL, "bad argument");
    if (n == 0) return SNAPHASH_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    uint64_t* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, 3 * n * sizeof(uint64_t)));
    uint64_t maxlen = 0;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i] & 7) { (void)hipFree(d); return fail(c, SNAPHASH_EINVAL, "offsets must be 8-byte aligned"); }
        maxlen = std::max(maxlen, lens[i]);
    }
    hipError_t e = hipMemcpyAsync(d, offsets, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, lens, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + 2 * n, file_index, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_fill_synthetic((uint8_t*)d_base, d, d + n, d + 2 * n, (uint32_t)n, maxlen, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, SNAPHASH_EDEVICE, std::string("fill_synthetic: ") + hipGetErrorString(e));
    return SNAPHASH_OK;
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

} // extern "C"
